#!/usr/bin/env python3
"""Summarise the --pmc passes of tools/profile_round.sh: python tools/pmc_ffn1.py gpurun_out/<tag> [configs...]
Per config: the FFN-1 GEMM's FETCH_SIZE / WRITE_SIZE per launch (FETCH doubled: gfx950 reports half the bytes of 16-B/lane
streaming reads, MI355X_MICROARCH.md, HBM), and matrix-pipe busy cycles per SIMD / kernel duration for the top kernels.
fp32 configs (1, 2, 3, 4, genea): FFN-1 is its own template instantiation (the gemm4 kernel launched L times per step with GELU).
Config 5 / 5b16 (fp16): every encoder GEMM is gemmh8b_kernel (+ a small-tile launch for the last rows); FFN-1 is the one dispatched
right after a LayerNorm and followed by another GEMM (FFN-2) rather than by attention."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
KB = 1024.0


def rows(path):
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def per_dispatch(rs, counter):
    d = {}
    for r in rs:
        if r["Counter_Name"] == counter:
            d[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return [d[k] for k in sorted(d)]


def ffn1_groups(seq, cfg):
    """index lists: the dispatches that make up one FFN-1 launch"""
    out = []
    if not cfg.startswith("5"):
        # QKV and FFN-1 both follow a LayerNorm; FFN-1 is followed by FFN-2 (a GEMM), QKV by attention
        for i, (n, _, _) in enumerate(seq):
            if "gemm4_kernel" in n and 0 < i < len(seq) - 1 and "layernorm" in seq[i - 1][0] and "gemm4_kernel" in seq[i + 1][0]:
                out.append([i])
        return out
    isg = lambda n: "gemmh8b" in n or "gemmh_kernel" in n        # noqa: E731
    for i, (n, _, _) in enumerate(seq):
        if isg(n) and i > 0 and "layernorm" in seq[i - 1][0]:
            g = [i]
            j = i + 1
            if "gemmh8b" in n and j < len(seq) and "gemmh_kernel" in seq[j][0]:
                g.append(j); j += 1                                # row-cut tail launch of the same GEMM (follows a 256 x 256 main only)
            if j < len(seq) and isg(seq[j][0]):                    # followed by FFN-2, not by attention (that would be QKV)
                out.append(g)
    return out


res = {}
for cfg in (sys.argv[2:] or ["2", "5"]):
    ent = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        seq = per_dispatch(rows(os.path.join(root, f"pmc_{ctr}_c{cfg}")), ctr)
        gs = ffn1_groups(seq, cfg)
        if gs:
            vals = [sum(seq[i][1] for i in g) for g in gs]
            ent[ctr + "_KB_avg"] = sum(vals) / len(vals)
            ent["launches_" + ctr] = len(vals)
            ent["kernel"] = " + ".join(sorted({seq[i][0][:60] for g in gs for i in g}))
    if "FETCH_SIZE_KB_avg" in ent and "WRITE_SIZE_KB_avg" in ent:
        ent["hbm_bytes_per_launch"] = int((2 * ent["FETCH_SIZE_KB_avg"] + ent["WRITE_SIZE_KB_avg"]) * KB)
    res[cfg] = ent
    seq = per_dispatch(rows(os.path.join(root, f"pmc_SQ_VALU_MFMA_BUSY_CYCLES_c{cfg}")), "SQ_VALU_MFMA_BUSY_CYCLES")
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for n, v, d in seq:
        a = agg[n[:70]]; a[0] += 1; a[1] += v; a[2] += d
    print(f"== config {cfg}: matrix-pipe busy cycles per SIMD (1024 SIMDs) / kernel nanosecond (= utilisation x shader clock in GHz)")
    for n, (c, v, d) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:8]:
        if v > 0:
            print(f"   {n:70s} calls={c:5d} avg_us={d / c / 1e3:8.1f} busy/SIMD/ns={v / 1024 / d:.3f}")
print(json.dumps(res, indent=1))
