#!/usr/bin/env python
"""gpurun_out/prof_*.json (written by tools/refresh_profiles.sh on the GPU box) -> tracked summaries under profiles/."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
TAG = sys.argv[1] if len(sys.argv) > 1 else "round1"
# C-ABI entry point -> kernel it launches (substring of the rocprofv3 kernel name)
ABI = {
    "nsim_lotd_gather_lm": "k_lotd_gather_lm<1, false>",   # round 3: f32 feature planes for the split-precision decoder
    "nsim_field_sdf": "k_field_sdf<2, 2, true",            # round 3: the sampling pass runs the split-precision decoder
    "nsim_field_fwd": "k_field<0, 2, 3",             # decoder half; its gather half is k_lotd_gather_lm<0, true>
    "nsim_field_fwd(gather)": "k_lotd_gather_lm<0, true>",
    "nsim_field_bwd_sdf": "k_field_bwd_j<0, 2",
    "nsim_field_bwd_rad": "k_rad_bwd_j<0>",
    "nsim_lotd_scatter": "k_lotd_scatter",
}


def main():
    bench = json.loads((G / "prof_bench.json").read_text().strip().splitlines()[-1])
    (P / f"{TAG}_bench_n1.json").write_text(json.dumps(bench, indent=1))
    st = json.loads((G / "prof_stats.json").read_text())
    steps = 48          # 16 warm-up + 32 timed
    lines = [f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity   (MI355X, {TAG})",
             f"{'kernel':72s} {'calls':>7s} {'us/step':>9s} {'avg us':>9s} {'%':>6s}"]
    for k in st["kernels"][:40]:
        lines.append(f"{k['name'][:72]:72s} {k['calls']:7d} {k['total_us'] / steps:9.1f} {k['avg_us']:9.2f} {k['pct']:6.2f}")
    (P / f"{TAG}_rocprofv3_kernel_stats.txt").write_text("\n".join(lines) + "\n")
    (P / f"{TAG}_rocprofv3_kernel_stats.json").write_text(json.dumps(dict(
        command="rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity",
        kernels=st["kernels"][:40], dispatch=st.get("dispatch", [])), indent=1))
    # the two kernels behind the nsim_field_fwd entry point, separately: bench.py states the forward decoder against its chain
    # ceiling on the decoder kernel's own time, from this recorded split
    def avg_us(sub):
        k = next((k for k in st["kernels"] if sub in k["name"]), None)
        return (k["avg_us"], k["calls"]) if k else (None, 0)
    (dec_us, dec_n), (gat_us, gat_n) = avg_us(ABI["nsim_field_fwd"]), avg_us(ABI["nsim_field_fwd(gather)"])
    if dec_us and gat_us:
        (P / "kernel_split.json").write_text(json.dumps(dict(
            nsim_field_fwd=dict(decoder_kernel=ABI["nsim_field_fwd"], decoder_avg_us=dec_us, gather_kernel=ABI["nsim_field_fwd(gather)"],
                                gather_avg_us=gat_us, launches=dec_n),
            _recorded=f"{TAG}: rocprofv3 --kernel-trace --stats of the bench command (profiles/{TAG}_rocprofv3_kernel_stats.txt)"), indent=1))
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
    import subprocess
    try:
        sha = subprocess.check_output(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        sha = "?"
    traffic["_recorded"] = f"{TAG} (tree after commit {sha})"
    (P / "traffic.json").write_text(json.dumps(traffic, indent=1))
    gp = G / "prof_gaps.json"
    if gp.exists():
        (P / f"{TAG}_gap_profile.json").write_text(gp.read_text())
    sq = G / "prof_pmc_sq.json"
    if sq.exists():
        q = json.loads(sq.read_text())
        disp = {d["name"]: d for d in q.get("dispatch", [])}
        ks = {}
        for name, c in q.get("pmc", {}).items():
            if "k_field" not in name and "k_rad" not in name and "k_lotd" not in name:
                continue
            v = {k: x["avg"] for k, x in c.items()}
            rec = dict(counters=v, launches=int(next(iter(c.values()))["n"]))
            if v.get("GRBM_GUI_ACTIVE"):
                rec["mfma_util_pct"] = round(100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 2)
            if v.get("SQ_WAVE_CYCLES"):
                w = v["SQ_WAVE_CYCLES"]
                rec["valu_active_per_wave_cycle"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / w, 3)
                rec["wait_any_frac"] = round(v.get("SQ_WAIT_ANY", 0.0) / w, 3)
                rec["wait_inst_any_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / w, 3)
                if v.get("SQ_BUSY_CU_CYCLES"):
                    rec["waves_per_cu_cycle"] = round(w / v["SQ_BUSY_CU_CYCLES"], 2)
            d = disp.get(name)
            if d:
                rec.update(vgpr=d["vgpr"], agpr=d["agpr"], lds=d["lds"], scratch=d["scratch"], avg_us=round(d["avg_ns"] / 1e3, 1))
            ks[name] = rec
        (P / f"{TAG}_rocprofv3_pmc_mfma.json").write_text(json.dumps(dict(
            command="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU "
                    "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 32 --warmup 16 "
                    "--no-cpu-baseline --no-variants --no-parity",
            note="per-launch averages. mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) "
                 "(gfx94x derived-metric formula); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; "
                 "waves_per_cu_cycle = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES = resident waves per busy CU (4 SIMDs).",
            kernels=ks), indent=1))
    # L2-side request counters (round 4: by hand; round 5: this block): the gathers against the L2 -> L1 request ceiling, the
    # scatter against the atomic-request ceiling -- bench.py quotes profiles/l2_requests.json as roofline.cache_ceilings
    l2 = G / "prof_pmc_l2.json"
    if l2.exists():
        q = json.loads(l2.read_text())
        disp = {d["name"]: d for d in q.get("dispatch", [])}
        CEIL_R, CEIL_A = 267e9, 20.7e9
        ks = {}
        for name, c in q.get("pmc", {}).items():
            if "k_lotd_gather_lm" not in name and "k_lotd_scatter" not in name:
                continue
            v = {k: x["avg"] for k, x in c.items()}
            d = disp.get(name)
            us = d["avg_ns"] / 1e3 if d else None
            rec = dict(launches=int(next(iter(c.values()))["n"]), avg_us_under_pmc=round(us, 1) if us else None,
                       tcp_tcc_read_req=int(v.get("TCP_TCC_READ_REQ_sum", 0)), tcp_cache_accesses=int(v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)),
                       atomic_req=int(v.get("TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", 0)), tcc_hit=int(v.get("TCC_HIT_sum", 0)),
                       tcc_miss=int(v.get("TCC_MISS_sum", 0)))
            if us:
                rec["read_req_per_s"] = round(rec["tcp_tcc_read_req"] / (us * 1e-6) / 1e9, 1)
                rec["frac_of_l2_read_req_ceiling"] = round(rec["tcp_tcc_read_req"] / (us * 1e-6) / CEIL_R, 3)
                if rec["atomic_req"]:
                    rec["atomic_req_per_s_G"] = round(rec["atomic_req"] / (us * 1e-6) / 1e9, 2)
                    rec["frac_of_atomic_req_ceiling"] = round(rec["atomic_req"] / (us * 1e-6) / CEIL_A, 3)
            ks[name] = rec
        rec_all = dict(
            command="rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum "
                    f"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity  (tools/refresh_profiles.sh; MI355X, {TAG}, tree after commit {sha})",
            calibration=dict(what="tools/gather_pair_bench: a random 4-byte gather from an L2-resident table sustains 256-267 G TCP->TCC "
                                  "read requests/s (profiles/round4_gather_pair_bench.txt) -- the L2 -> L1 request ceiling; tools/atomic_bench4 / "
                                  "tools/atomic_alloc_bench: 20.7-21.0 G 16-byte atomic requests/s whatever the scope, table size, XCD partition "
                                  "or allocation (profiles/round4_atomic_scope_probe.txt, profiles/round5_atomic_alloc_bench.txt)",
                             l2_read_req_ceiling_per_s=CEIL_R, atomic_req_ceiling_per_s=CEIL_A),
            kernels=ks)
        (P / f"{TAG}_l2_requests.json").write_text(json.dumps(rec_all, indent=1))
        (P / "l2_requests.json").write_text(json.dumps(dict(rec_all, _recorded=f"{TAG} (tree after commit {sha})"), indent=1))
    # round 3: the distant-model step, the street configuration, per-level scatter timing
    def table(src, steps, title, dst, own_only=False):
        f = G / src
        if not f.exists():
            return
        d = json.loads(f.read_text())
        out_l = [title, f"{'kernel':72s} {'calls':>7s} {'us/step':>9s} {'avg us':>9s} {'%':>6s}"]
        rows = d.get("kernels", [])
        if own_only:        # the library's kernels only (k_* / _Z6k_adam...): ATen rows of such a run are set-up work
            rows = [k for k in rows if re.search(r"(^|\s|::)k_[a-z0-9_]+|_Z\d+k_", k["name"]) and "at::native" not in k["name"]]
            tot = sum(k["total_us"] for k in rows)
            out_l.insert(1, f"library kernels only: {tot / steps:.1f} us per step in {sum(k['calls'] for k in rows) / steps:.0f} launches "
                            f"(the '%' column is of the WHOLE process, set-up included)")
        for k in rows[:32]:
            out_l.append(f"{k['name'][:72]:72s} {k['calls']:7d} {k['total_us'] / steps:9.1f} {k['avg_us']:9.2f} {k['pct']:6.2f}")
        (P / dst).write_text("\n".join(out_l) + "\n")
    table("prof_distant_stats.json", 24, f"rocprofv3 --kernel-trace --stats -- python bench.py --distant --steps 16 --warmup 8 "
          f"--no-cpu-baseline --no-variants --no-parity   (MI355X, {TAG}; 24 steps)", f"{TAG}_distant_rocprofv3_kernel_stats.txt")
    table("prof_street_stats.json", 12, f"rocprofv3 --kernel-trace --stats -- python bench.py --config street --steps 8 --warmup 4   "
          f"(MI355X, {TAG}; 12 steps; ATen kernels left out: they are dominated by the one-off dataset synthesis of the set-up)",
          f"{TAG}_street_rocprofv3_kernel_stats.txt", own_only=True)
    for src, dst in (("prof_distant_bench.json", f"{TAG}_bench_n1_distant.json"), ("prof_street_bench.json", f"{TAG}_bench_n1_street.json"),
                     ("prof_scatter_levels.json", f"{TAG}_scatter_levels.json"), ("prof_distant_pmc_sq.json", f"{TAG}_distant_pmc_sq.json"),
                     ("prof_distant_fusedgather.json", f"{TAG}_bench_n1_distant_fused_gather.json")):
        f = G / src
        if f.exists():
            txt = f.read_text().strip()
            try:
                (P / dst).write_text(json.dumps(json.loads(txt.splitlines()[-1] if src.endswith("bench.json") or "fusedgather" in src else txt), indent=1))
            except Exception:
                (P / dst).write_text(txt)
    for src, dst in (("prof_ktime.txt", f"{TAG}_ktime_timelines.txt"), ("prof_field_bench.jsonl", f"{TAG}_field_bench.jsonl"),
                     ("prof_bench_noreplicas.json", f"{TAG}_bench_n1_no_grad_replicas.json"),
                     ("prof_distant_lmgather.json", f"{TAG}_bench_n1_distant_noprofiler.json")):
        f = G / src
        if f.exists() and f.stat().st_size:
            txt = f.read_text()
            if src.endswith(".json"):
                try:
                    txt = json.dumps(json.loads(txt.strip().splitlines()[-1]), indent=1)
                except Exception:
                    pass
            (P / dst).write_text(txt)
    print(json.dumps(traffic, indent=1))
    print(bench["value"], bench["ms_per_step"], bench["roofline"])


if __name__ == "__main__":
    main()
