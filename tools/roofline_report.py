#!/usr/bin/env python3
"""Per-kernel roofline table of one train step (profiles/rNN_roofline_per_kernel.csv + rNN_pmc_summary.txt).

Inputs (all produced on the GPU box by tools/profile_round.sh):
  --stats   rocprofv3 --kernel-trace --stats CSV of the single-stream bench run (kernels run alone)
  --steps   number of train steps in that run (warm-up + timed + the roofline leg, if any)
  --algo    JSON written by tools/profile_step.py --json: algorithmic FLOPs / bytes per GEMM kernel (HIP-event timed)
  --traffic JSON of tools/pmc_traffic.py (FETCH_SIZE x2 + WRITE_SIZE per launch, separate --pmc passes)
  --sq      counter_collection CSVs of the SQ passes (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_LDS_BANK_CONFLICT,
            SQ_LDS_IDX_ACTIVE, GRBM_GUI_ACTIVE ...), any number of files
Roofs (/opt/skills/guides/MI355X_MICROARCH.md): HBM 8000 GB/s, fp32 MFMA 157.3 TFLOP/s, dense bf16 MFMA 2500 TFLOP/s.
A kernel is priced against the matrix roof with the FLOPs it ISSUES (Winograd: direct x 16/36; six-product tiles: x 6
bf16 MFMAs per multiply, three-fp16-product tiles -- rocprof names "<.., true>" -- x 3, direct split tile + 1/9 for its zero tap)
and against HBM with its measured PMC bytes.  Round 6: every row also carries `frac_algorithmic` = ALGORITHMIC
bytes per second / 8 TB/s (over-fetch is not achievement) and `waste` = PMC bytes / algorithmic bytes."""
import argparse
import collections
import csv
import json
import re

HBM, FP32, BF16 = 8000.0, 157.3, 2500.0


def norm(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", k).strip()          # argument list
    return k


def algo_key(k):
    """rocprof kernel name -> name used by refid_amd.ops.PROFILE (tools/profile_step.py)."""
    if k.startswith("conv_wino6_kernel"):
        m = re.match(r"conv_wino6_kernel<(\d+)(?:, (true|false))?>", k)           # <NT, F16>: ops.PROFILE names "<NT>" / "<NT, true>"
        if not m: return "conv_wino6_kernel<2>"
        return f"conv_wino6_kernel<{m.group(1)}, true>" if m.group(2) == "true" else f"conv_wino6_kernel<{m.group(1)}>"
    if k.startswith("conv_wino_kernel"): return k
    if k.startswith("wgrad_wino_kernel"): return "wgrad_wino_kernel"
    if k.startswith("wgrad_wino24_kernel") and "true" in k: return "wgrad_wino24_down_kernel"
    if k.startswith("wgrad_wino24_kernel"): return "wgrad_wino24_kernel"
    if k.startswith("conv_pw_kernel"):
        m = re.match(r"conv_pw_kernel<(\d+), (\d+)", k)
        return f"conv_pw_kernel<{m.group(1)}, {m.group(2)}>"
    if k.startswith("conv_split_kernel"):                                          # <MT, NT, PL, KS, MODE, F16>
        return "conv_split_kernel<3, fp16>" if k.rstrip(">").endswith("true") else "conv_split_kernel<6>"
    m = re.match(r"wgrad_kernel<WCfg<(\d+), (\d+), (\d+)", k)
    if m: return f"wgrad_kernel<{m.group(1)}x{m.group(2)}s{m.group(3)}>"
    if k.startswith("wgrad_pw_kernel") or k.startswith("wgrad_pws_kernel"): return "wgrad_kernel<1x1s1>"
    m = re.match(r"conv_igemm_kernel<(Cfg<[^>]*>)", k)
    if m: return f"conv_igemm_kernel<{m.group(1)}>"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True); ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--algo"); ap.add_argument("--traffic"); ap.add_argument("--sq", nargs="*", default=[])
    ap.add_argument("--out", required=True); ap.add_argument("--summary")
    a = ap.parse_args()
    algo = json.load(open(a.algo)) if a.algo else {}
    traffic = json.load(open(a.traffic)) if a.traffic else {}
    sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in a.sq:
        for r in csv.DictReader(open(path)):
            v = sq[norm(r["Kernel_Name"])][r["Counter_Name"]]
            v[0] += float(r["Counter_Value"]); v[1] += 1
    rows = []
    total_ns = 0.0
    stats = list(csv.DictReader(open(a.stats)))
    for r in stats:
        total_ns += float(r["TotalDurationNs"])
    for r in stats:
        k = norm(r["Name"])
        calls, tot, avg = int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])
        ms_step = tot / a.steps / 1e6
        if ms_step < 0.3:
            continue
        ak = algo_key(k)
        al = algo.get(ak) if ak else None
        tr = traffic.get(k)
        pmc_bytes = tr["hbm_bytes_per_launch"] if tr else None
        issued = bound = frac = rate = unit = None
        alg_bytes = None
        if al and al["launches"]:
            # several rocprof kernels can share one PROFILE name (conv_pw true/false, split variants): scale by time share
            fl_per_s = al["flops"] / (al["ms"] * 1e-3)                  # direct-conv FLOP/s over that family
            alg_bytes = al["bytes"] / al["launches"]
            mult, peak, bound = 1.0, FP32, "mfma-fp32"
            if "wino6" in k and "true" in k: mult, peak, bound = 16.0 / 36.0 * 3.0, BF16, "mfma-fp16"
            elif "wino6" in k: mult, peak, bound = 16.0 / 36.0 * 6.0, BF16, "mfma-bf16"
            elif "wino24" in k and "true" in k: mult = 12.0 / 16.0
            elif "wino24" in k: mult = 12.0 / 36.0
            elif "wino" in k: mult = 16.0 / 36.0
            elif "split" in k and k.rstrip(">").endswith("true"): mult, peak, bound = 3.0, BF16, "mfma-fp16"
            elif "split" in k: mult, peak, bound = 6.0 * 10.0 / 9.0 if "3, 2, 0" in k else 6.0, BF16, "mfma-bf16"
            issued = fl_per_s * mult / 1e12
            rate, unit, frac = issued, "TFLOP/s issued", issued / peak
            if pmc_bytes and pmc_bytes / (avg * 1e-9) / 1e9 / HBM > frac:
                bound = "hbm"
        if pmc_bytes and (bound in (None, "hbm")):
            gbs = pmc_bytes / (avg * 1e-9) / 1e9
            bound, rate, unit, frac = "hbm", gbs, "GB/s (PMC bytes)", gbs / HBM
        c = sq.get(k, {})
        mean = lambda n: (c[n][0] / c[n][1]) if n in c and c[n][1] else None         # noqa: E731
        mfma, busy = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_BUSY_CU_CYCLES")
        gui = mean("GRBM_GUI_ACTIVE")
        ldsc, ldsa = mean("SQ_LDS_BANK_CONFLICT"), mean("SQ_LDS_IDX_ACTIVE")
        mfma_util = mfma / (gui / 8.0 * 1024.0) if mfma and gui else None    # 1024 SIMDs; GUI_ACTIVE summed over 8 XCDs
        rows.append(dict(kernel=k, launches_per_step=round(calls / a.steps, 1), ms_per_step=round(ms_step, 2),
                         share=round(tot / total_ns, 4), avg_us=round(avg / 1e3, 1), bound=bound,
                         rate=None if rate is None else round(rate, 1), unit=unit, frac=None if frac is None else round(frac, 3),
                         pmc_hbm_bytes_per_launch=pmc_bytes, algorithmic_bytes_per_launch=None if alg_bytes is None else round(alg_bytes),
                         frac_algorithmic=None if alg_bytes is None else round(alg_bytes / (avg * 1e-9) / 1e9 / HBM, 3),
                         waste=None if not (alg_bytes and pmc_bytes) else round(pmc_bytes / alg_bytes, 2),
                         mfma_busy_over_busy=None if mfma_util is None else round(mfma_util, 3),
                         lds_conflict_share=None if not (ldsc is not None and ldsa) else round(ldsc / ldsa, 3)))
    rows.sort(key=lambda r: -r["ms_per_step"])
    with open(a.out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
    if a.summary:
        with open(a.summary, "w") as f:
            f.write("PMC summary per kernel (per-launch means; rocprofv3 --pmc in passes of their own, --kernel-trace only).\n"
                    "MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); LDS conflict share = "
                    "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n")
            for k, c in sorted(sq.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
                if not any(r["kernel"] == k for r in rows):
                    continue
                f.write(k + "\n")
                for n, v in sorted(c.items()):
                    f.write(f"    {n:32s} {v[0] / v[1]:18.1f}  (n={v[1]})\n")
    print(f"{len(rows)} kernels >= 0.3 ms/step -> {a.out}")
    for r in rows[:14]:
        print(f"  {r['kernel'][:58]:58s} {r['ms_per_step']:7.2f} ms  {str(r['bound']):10s} frac {r['frac']}  hbm(alg) {r['frac_algorithmic']}  "
              f"waste {r['waste']}  mfma {r['mfma_busy_over_busy']}")


if __name__ == "__main__":
    main()
