// misc.hip -- version / error strings of the C ABI.
#include "nsim_common.h"

extern "C" {

int nsim_version(void) { return 100; }

const char* nsim_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case 2: return "negative size";
    case 3: return "bad channel count / op code";
    case 4: return "required output pointer is NULL";
    case 5: return "missing occupancy/AABB meta or non-positive step";
    case 10: return "LoTD meta is NULL";
    case 11: return "LoTD n_feats must be 2";
    case 12: return "LoTD num_levels out of range";
    case 13: return "LoTD level resolution < 2";
    case 14: return "LoTD dense level size != res^3";
    case 15: return "LoTD unknown level type";
    case 16: return "LoTD level offset must be even";
    case 17: return "LoTD hash table size must be a power of two";
    case 29: return "per-ray instance offsets (batched model) need ridx";
    case 28: return "h / dh/dx planes (and dh / g hand-off planes when dgrid is requested) are required";
    case 27: return "radiance backward needs the saved forward nablas / rgb and a [S,3] scratch buffer";
    case 30: return "sky meta is NULL";
    case 31: return "sky input width 3 + 6 n_frequencies + n_appear must be <= 96";
    case 32: return "sky model with n_appear > 0 needs h_appear";
    case 20: return "field meta is NULL";
    case 21: return "field kernels take 1..32 LoTD levels (<= 64 input features)";
    case 33: return "pyramids with more than 16 levels exist on the level-major path only: the planes arguments are required";
    case 40: return "permuto meta is NULL";
    case 41: return "permuto in_dim must be 2..8 (>= 3 for the field front end)";
    case 42: return "permuto num_levels must be 1..32";
    case 43: return "permuto n_feats must be 2";
    case 44: return "permuto hashmap_size must be a power of two";
    case 34: return "too many (device, stream) pairs with a registered gradient scratch (64)";
    case 22: return "sdf_D must be 1 or 2";
    case 23: return "precision must be 0 (fp16 MFMA) or 1 (f32 MFMA)";
    case 24: return "need either x or (rays_o, rays_d, t, ridx)";
    case 25: return "radiance needs rays_d and ridx";
    case 26: return "gradient output pointer is NULL";
    case 37: return "compose collect: at most 64 sources";
    case 36: return "wide decoder: 0..10 embedding frequencies and at most 128 first-layer inputs (2 num_levels + 3 + 6 n_freq)";
    default: return code >= 1000 ? "HIP launch error (code - 1000 = hipError_t)" : "unknown error";
  }
}

}  // extern "C"
