#!/usr/bin/env python3
"""How busy is the GPU during a train step -- WITHOUT a tracer in the timed run.

rocprofv3's kernel trace stretches the gaps between ~5000 short launches (round 3: "idle 55.9 %" under the tracer), but it
does not change kernel DURATIONS.  So: busy = (sum of kernel durations of one single-stream step, from the traced run's
--stats CSV) / (wall time per step of the SAME single-stream configuration measured untraced by bench.py).  With every
kernel on one stream nothing overlaps, so that quotient is the fraction of the step the GPU executes kernels; the rest is
launch gaps / host enqueue.  The default (multi-stream) step is reported beside it: it can only be shorter than the
single-stream one by what the side streams overlap.

  python tools/busy_report.py --stats <nooverlap_kernel_stats.csv> --stat-steps 4 \
         --single <untraced single-stream bench json> --default <untraced default bench json>
"""
import argparse
import csv
import json


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--stat-steps", type=int, required=True, help="train steps in the traced run (warm-up + timed + roofline leg)")
    ap.add_argument("--single", required=True)
    ap.add_argument("--default", required=True)
    a = ap.parse_args()
    tot = sum(int(r["TotalDurationNs"]) for r in csv.DictReader(open(a.stats)))
    launches = sum(int(r["Calls"]) for r in csv.DictReader(open(a.stats)))
    k_ms = tot / 1e6 / a.stat_steps
    s, d = last_json(a.single), last_json(a.default)
    print(f"kernel time per step (sum of durations, traced single-stream run / {a.stat_steps} steps): {k_ms:.1f} ms, "
          f"{launches // a.stat_steps} launches")
    print(f"single-stream step, untraced (REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0): {s['ms_per_step']:.1f} ms "
          f"-> GPU busy {100 * k_ms / s['ms_per_step']:.1f} %, gaps {s['ms_per_step'] - k_ms:.1f} ms")
    print(f"default step, untraced (the shipped stream layout): {d['ms_per_step']:.1f} ms "
          f"= {100 * d['ms_per_step'] / k_ms:.1f} % of the serial kernel time")


if __name__ == "__main__":
    main()
