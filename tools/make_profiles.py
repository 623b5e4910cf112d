#!/usr/bin/env python
"""gpurun_out/prof_*.json (written by tools/refresh_profiles.sh on the GPU box) -> tracked summaries under profiles/."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
TAG = sys.argv[1] if len(sys.argv) > 1 else "round1"
# C-ABI entry point -> kernel it launches (substring of the rocprofv3 kernel name)
ABI = {
    "nsim_lotd_gather_lm": "k_lotd_gather_lm<0, false>",
    "nsim_field_sdf": "k_field_sdf<0, 2, true>",
    "nsim_field_fwd": "k_field<0, 2, 3>",            # decoder half; its gather half is k_lotd_gather_lm<0, true>
    "nsim_field_fwd(gather)": "k_lotd_gather_lm<0, true>",
    "nsim_field_bwd_sdf": "k_field<0, 2, 2>",
    "nsim_field_bwd_rad": "k_rad_bwd<0>",
    "nsim_lotd_scatter": "k_lotd_scatter",
}


def main():
    bench = json.loads((G / "prof_bench.json").read_text().strip().splitlines()[-1])
    (P / f"{TAG}_bench_n1.json").write_text(json.dumps(bench, indent=1))
    st = json.loads((G / "prof_stats.json").read_text())
    steps = 48
    lines = [f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 16 --no-cpu-baseline   (MI355X, {TAG})",
             f"{'kernel':72s} {'calls':>7s} {'us/step':>9s} {'avg us':>9s} {'%':>6s}"]
    for k in st["kernels"][:40]:
        lines.append(f"{k['name'][:72]:72s} {k['calls']:7d} {k['total_us'] / steps:9.1f} {k['avg_us']:9.2f} {k['pct']:6.2f}")
    (P / f"{TAG}_rocprofv3_kernel_stats.txt").write_text("\n".join(lines) + "\n")
    (P / f"{TAG}_rocprofv3_kernel_stats.json").write_text(json.dumps(dict(
        command="rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 16 --no-cpu-baseline",
        kernels=st["kernels"][:40], dispatch=st.get("dispatch", [])), indent=1))
    fe = json.loads((G / "prof_pmc_fetch.json").read_text()).get("pmc", {})
    wr = json.loads((G / "prof_pmc_write.json").read_text()).get("pmc", {})
    out, traffic = {}, {}
    for abi, sub in ABI.items():
        kf = next((k for k in fe if sub in k), None)
        kw = next((k for k in wr if sub in k), None)
        if kf is None or kw is None:
            continue
        f, w = fe[kf]["FETCH_SIZE"], wr[kw]["WRITE_SIZE"]
        out[abi] = dict(kernel=kf, FETCH_SIZE_KB_avg_per_launch=f["avg"], WRITE_SIZE_KB_avg_per_launch=w["avg"],
                        launches=f["n"])
        traffic[abi] = int((f["avg"] + w["avg"]) * 1024)
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python bench.py --steps 32 "
            "--warmup 16 --no-cpu-baseline`, MI355X; traffic.json = (FETCH_SIZE + WRITE_SIZE) KB x 1024, averaged over the "
            "launches of each kernel (small 4096-point launches included). Raw counters: the gfx950 x2 correction of "
            "MI355X_MICROARCH.md applies to wide coalesced streams only and is NOT applied here -- the gather / atomic "
            "access widths of these kernels are uncalibrated. Counters are memory-side (Infinity-Cache hits included).")
    (P / f"{TAG}_rocprofv3_pmc_hbm.json").write_text(json.dumps(dict(note=note, kernels=out), indent=1))
    (P / "traffic.json").write_text(json.dumps(traffic, indent=1))
    print(json.dumps(traffic, indent=1))
    print(bench["value"], bench["ms_per_step"], bench["roofline"])


if __name__ == "__main__":
    main()
