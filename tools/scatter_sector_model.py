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


def level_layout(lod_res, T):
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


def distinct_per_row(a):
    """a [rows, n] int64 with -1 = inactive -> number of distinct non-negative values per row"""
    s = np.sort(a, axis=1)
    first = np.concatenate([np.ones((s.shape[0], 1), bool), s[:, 1:] != s[:, :-1]], 1)
    return int((first & (s >= 0)).sum())


def model(x, ridx, lod_res, T, group="strided", dedup=True, chunk=64):
    """-> per level dict(sectors, instr, atomics)   group: 'strided' (lanes 4q+I per issue: the round 1-5 kernel) |
    'consecutive' (lanes 16I+q per issue)"""
    S = x.shape[0]
    pad = (-S) % chunk
    sizes, types, offs = level_layout(lod_res, T)
    out = []
    u = x * 0.5 + 0.5
    for l, R in enumerate(lod_res):
        R = int(R)
        pos = u * (R - 1)
        c0 = np.clip(np.floor(pos), 0, R - 2).astype(np.int64)
        tot_sec = tot_instr = tot_atom = 0
        for yz in range(4):
            sec_dx, emit_dx = [], []
            for dx in range(2):
                cx, cy, cz = c0[:, 0] + dx, c0[:, 1] + (yz & 1), c0[:, 2] + (yz >> 1)
                if types[l] == "dense":
                    idx = cx + R * (cy + R * cz)
                else:
                    h = cx.astype(np.uint32) ^ (cy.astype(np.uint32) * P1) ^ (cz.astype(np.uint32) * P2)
                    idx = (h & np.uint32(T - 1)).astype(np.int64)
                idxp = np.concatenate([idx, np.full(pad, -1)]).reshape(-1, chunk)
                if dedup:
                    nxt = np.concatenate([idxp[:, 1:], np.full((idxp.shape[0], 1), -2)], 1)
                    emit = (idxp != nxt) & (idxp >= 0)
                else:
                    emit = idxp >= 0
                sector = (offs[l] + 2 * idxp) // 16
                sec_dx.append(np.where(emit, sector, -1))
                emit_dx.append(emit)
                tot_atom += 2 * int(emit.sum())
            both = np.stack(sec_dx, -1)                      # [chunks, 64, 2]
            nc = both.shape[0]
            if group == "strided":
                rows = both.reshape(nc, 16, 4, 2).transpose(0, 2, 1, 3).reshape(nc * 4, 32)      # issue I = lanes 4q + I
            else:
                rows = both.reshape(nc * 4, 32)                                                   # issue I = lanes 16I + q
            tot_sec += distinct_per_row(rows)
            tot_instr += int((rows >= 0).any(1).sum())
        out.append(dict(level=l, res=R, type=types[l], sectors=tot_sec, instr=tot_instr, lane_atomics=tot_atom,
                        sectors_per_point=round(tot_sec / S, 3)))
    return out


def main():
    d = np.load(sys.argv[1])
    if "x" in d.files:
        x = d["x"]
    else:
        x = d["o"][d["ridx"]] + d["t"][:, None] * d["d"][d["ridx"]]
    ridx = d["ridx"]
    lod_res, T = [int(r) for r in d["lod_res"]], int(d["hashmap_size"])
    aabb = d["aabb"]
    x = (x - aabb[0]) / (aabb[1] - aabb[0]) * 2 - 1
    res = dict(points=int(x.shape[0]), variants={})
    for name, kw in (("round5_strided_dedup", dict(group="strided", dedup=True)),
                     ("consecutive_dedup", dict(group="consecutive", dedup=True)),
                     ("consecutive_nodedup", dict(group="consecutive", dedup=False)),
                     ("strided_nodedup", dict(group="strided", dedup=False))):
        lv = model(x, ridx, lod_res, T, **kw)
        tot = sum(v["sectors"] for v in lv)
        res["variants"][name] = dict(sectors=tot, sectors_per_point=round(tot / x.shape[0], 2),
                                     us_at_21G=round(tot / 21.0e3, 1), instr=sum(v["instr"] for v in lv), levels=lv)
        print(name, "sectors/point", round(tot / x.shape[0], 2), "-> us at 21 G/s:", round(tot / 21.0e3, 1),
              [v["sectors_per_point"] for v in lv])
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
