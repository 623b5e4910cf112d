#!/usr/bin/env python
"""Host model of the table scatter's atomic traffic: distinct 64-byte sectors per wave instruction, summed over a launch.

Why a model: the MI355X retires float atomics at ~21 G (distinct 64-byte sector, wave instruction) pairs per second, whatever
the number of lanes of the instruction inside the sector (profiles/round4_atomic_line_bench.txt,
profiles/round6_atomic_conflict_bench.txt), and k_lotd_scatter runs at 0.91 of that rate -- so its time is the number of such
pairs, a pure function of the sample positions and of the lane -> sample mapping.  This script counts them for the kernel's
mapping and for candidate mappings on a sample set dumped from the bench step (tools/dump_scatter_points.py), per level.

usage: python tools/scatter_sector_model.py gpurun_out/scatter_points_object.npz [out.json]
"""
import json
import sys

import numpy as np

P1, P2 = np.uint32(2654435761), np.uint32(805459861)


def level_layout(lod_res, T):      # (cubic levels; model() lays cuboid ones out itself)
    sizes, types = [], []
    for r in lod_res:
        n = int(r) ** 3
        if n <= T:
            sizes.append(n); types.append("dense")
        else:
            sizes.append(T); types.append("hash")
    offs, o = [], 0
    for s in sizes:
        offs.append(o)
        o += 2 * s
        o += o & 1
    return sizes, types, offs


def requests_per_row(rows, sector_of):
    """rows [n, m] int64 table-entry indices of the lanes of one atomic instruction (-1 = inactive) -> atomic requests:
    one per distinct 64-byte sector, and lanes adding to the SAME address ride in separate requests
    (profiles/round6_atomic_conflict_bench.txt: K quads on the same 16 bytes cost K requests; K quads on K different 16-byte
    pieces of one sector cost one) -> per sector, the largest multiplicity of any of its addresses."""
    s = np.sort(rows, axis=1)
    valid = s >= 0
    m = s.shape[1]
    same = np.concatenate([np.zeros((s.shape[0], 1), bool), s[:, 1:] == s[:, :-1]], 1) & valid
    mult = np.ones_like(s)
    for k in range(1, m):
        mult[:, k] = np.where(same[:, k], mult[:, k - 1] + 1, 1)
    sec = np.where(valid, sector_of(s), -1)
    newsec = np.concatenate([np.ones((s.shape[0], 1), bool), sec[:, 1:] != sec[:, :-1]], 1)
    rm = mult.copy()
    for k in range(1, m):
        rm[:, k] = np.where(newsec[:, k], mult[:, k], np.maximum(rm[:, k - 1], mult[:, k]))
    last = np.concatenate([newsec[:, 1:], np.ones((s.shape[0], 1), bool)], 1)
    return int((rm * (last & valid)).sum())


def model(u, lod_res, T, group="strided", dedup=True, cross=False, slots="offset", entry_bytes=8, chunk=64):
    """-> per level dict(requests, ...).   u [S, 3]: positions in the pyramid's unit cube; lod_res [L] or [L, 3] (cuboid levels).
    slots: 'offset' -- slot (dx, yz) holds the corner c0 + (dx, yz & 1, yz >> 1) (the kernel of rounds 1-5) | 'parity' -- slot
    (dx, yz) holds the vertex whose coordinates have those parities (round 6: a vertex shared by neighbouring cells keeps its slot,
    so the per-slot run merge folds it);  group: 'strided' (issue I = lanes 4q + I) | 'consecutive' (issue I = lanes 16 I + q:
    NSIM_SCATTER_GROUP=1);  dedup: runs of equal vertex index on consecutive lanes of a slot are summed by shuffles first (the
    kernel does);  cross ('offset' slots only): fold a sample's x + 1 corner into the next sample's x corner when they are the same
    vertex;  entry_bytes: 8 = two f32 per vertex (the kernel), 4 = a packed 2-byte pair per vertex (global_atomic_pk_add_*)."""
    S = u.shape[0]
    pad = (-S) % chunk
    res = np.asarray(lod_res)
    if res.ndim == 1:
        res = np.stack([res] * 3, -1)
    sizes, types, offs, o = [], [], [], 0
    for r in res:
        n = int(r[0]) * int(r[1]) * int(r[2])
        sizes.append(n if n <= T else T)
        types.append("dense" if n <= T else "hash")
        offs.append(o)
        o += 2 * sizes[-1]
    per_sector = 64 // entry_bytes
    out = []
    for l, R in enumerate(res):
        R = R.astype(np.int64)
        pos = u * (R - 1)
        c0 = np.minimum(np.maximum(np.floor(pos), 0), R - 2).astype(np.int64)
        par = c0 & 1 if slots == "parity" else np.zeros_like(c0)
        tot_req = tot_atom = 0
        base = offs[l] // 2
        for yz in range(4):
            idxs, emits = [], []
            for dx in range(2):
                cx = c0[:, 0] + (dx ^ par[:, 0])
                cy = c0[:, 1] + ((yz & 1) ^ par[:, 1])
                cz = c0[:, 2] + ((yz >> 1) ^ par[:, 2])
                if types[l] == "dense":
                    idx = cx + R[0] * (cy + R[1] * cz)
                else:
                    h = cx.astype(np.uint32) ^ (cy.astype(np.uint32) * P1) ^ (cz.astype(np.uint32) * P2)
                    idx = (h & np.uint32(T - 1)).astype(np.int64)
                idxp = np.concatenate([idx, np.full(pad, -1)]).reshape(-1, chunk)
                if dedup:
                    nxt = np.concatenate([idxp[:, 1:], np.full((idxp.shape[0], 1), -2)], 1)
                    emit = (idxp != nxt) & (idxp >= 0)
                else:
                    emit = idxp >= 0
                idxs.append(idxp)
                emits.append(emit)
            if cross and slots == "offset":
                nxt0 = np.concatenate([idxs[0][:, 1:], np.full((idxs[0].shape[0], 1), -2)], 1)
                emits[1] = emits[1] & ~(idxs[1] == nxt0)
            both = np.stack([np.where(emits[0], idxs[0], -1), np.where(emits[1], idxs[1], -1)], -1)      # [chunks, 64, 2]
            nc = both.shape[0]
            if group == "strided":
                rows = both.reshape(nc, 16, 4, 2).transpose(0, 2, 1, 3).reshape(nc * 4, 32)
            else:
                rows = both.reshape(nc * 4, 32)
            tot_req += requests_per_row(rows, lambda e: (base + e) // per_sector)
            tot_atom += 2 * int(emits[0].sum() + emits[1].sum())
        out.append(dict(level=l, res=[int(v) for v in R], type=types[l], requests=tot_req, lane_atomics=tot_atom,
                        requests_per_point=round(tot_req / S, 3)))
    return out


def main():
    d = np.load(sys.argv[1])
    x = d["x"] if "x" in d.files else d["o"][d["ridx"]] + d["t"][:, None] * d["d"][d["ridx"]]
    lod_res, T = np.asarray(d["lod_res"]), int(d["hashmap_size"])
    aabb = d["aabb"]
    u = (x - aabb[0]) / (aabb[1] - aabb[0])
    res = dict(points=int(x.shape[0]), levels=[[int(v) for v in np.atleast_1d(r)] for r in lod_res], hashmap_size=T,
               rule="requests = per wave instruction, per distinct 64-byte sector: the largest number of lanes on one address "
                    "(tools/atomic_conflict_bench.hip); ~21 G requests/s chip-wide", variants={})
    for name, kw in (("rounds 1-5 kernel: corner-offset slots, strided issue, run dedup", dict()),
                     ("corner-offset slots, no dedup", dict(dedup=False)),
                     ("corner-offset slots, consecutive issue", dict(group="consecutive")),
                     ("corner-offset slots, consecutive issue + cross-corner fold", dict(group="consecutive", cross=True)),
                     ("corner-offset slots, packed 2-byte pair per vertex (pk_add)", dict(entry_bytes=4)),
                     ("PARITY slots, strided issue (round 6 kernel, default)", dict(slots="parity")),
                     ("PARITY slots, consecutive issue (NSIM_SCATTER_GROUP=1)", dict(slots="parity", group="consecutive")),
                     ("PARITY slots, consecutive issue, packed 2-byte pair per vertex", dict(slots="parity", group="consecutive", entry_bytes=4))):
        lv = model(u, lod_res, T, **kw)
        tot = sum(v["requests"] for v in lv)
        res["variants"][name] = dict(requests=tot, requests_per_point=round(tot / x.shape[0], 2), us_at_21G=round(tot / 21.0e3, 1),
                                     per_level=[v["requests_per_point"] for v in lv])
        print(f"{name:72s} requests/point {tot / x.shape[0]:6.2f}  -> {tot / 21.0e3:7.1f} us at 21 G/s")
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
