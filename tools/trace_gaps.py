"""Where a timed train step's wall time goes on the GPU: kernel durations and the idle gaps between consecutive kernels,
from a rocprofv3 --kernel-trace CSV of a bench.py run.
usage: python tools/trace_gaps.py <kernel_trace.csv> [steps_to_show]
The trace is cut into steps at the optimiser launch that opens a pipelined step (sgd_range_kernel); the summary is over the
steps of the LAST timed region that has at least 8 of them (the headline arithmetic runs first, the others after it - pass
--first to summarise the first region instead)."""
import csv
import statistics
import sys


def short(name):
    for key, s in (("sgd_range", "sgd_open"), ("gemm_tiles", "gemm"), ("heads_kernel", "heads"), ("sgd_kernel", "sgd"),
                   ("set_hyper", "set_hyper"), ("to_bf16", "to_bf16"), ("grad_norm", "grad_norm"), ("train_steps_kernel", "steps")):
        if key in name:
            if s == "gemm":
                a = name[name.index("<") + 1:name.index(">")].replace(" ", "")
                return "gemm<" + a + ">"
            return s
    return name[:40]


def main():
    path = sys.argv[1]
    first = "--first" in sys.argv
    nshow = int(next((a for a in sys.argv[2:] if a.isdigit()), "2"))
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # regions of back-to-back steps: split where the gap exceeds 1 ms (fences, engine construction, CPU baseline)
    regions, cur = [], []
    for i, (s, e, n) in enumerate(rows):
        if cur and s - cur[-1][1] > 1_000_000:
            regions.append(cur)
            cur = []
        cur.append((s, e, n))
    if cur:
        regions.append(cur)
    steps_of = []
    for reg in regions:
        idx = [i for i, k in enumerate(reg) if k[2] == "sgd_open"]
        steps = [reg[a:b] for a, b in zip(idx, idx[1:]) if b - a >= 5]      # (not the measurement aids' repeated single launches)
        if len(steps) >= 8:
            steps_of.append(steps)
    if not steps_of:
        print("no region with >= 8 pipelined steps found")
        return
    steps = steps_of[0] if first else steps_of[-1]
    # the timed region of bench.py is the last `steps` steps of a region (warmup precedes it without a gap > 1 ms)
    wall = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps, steps[1:])]
    busy = [sum(e - s for s, e, _ in st) / 1e3 for st in steps]
    print(f"{len(steps)} steps in the region; step wall (start to start) us: median {statistics.median(wall):.1f}  mean {statistics.mean(wall):.1f}  "
          f"min {min(wall):.1f}  max {max(wall):.1f}")
    print("step walls in order, us: " + " ".join(f"{w:.0f}" for w in wall))
    print(f"kernel-busy us per step: median {statistics.median(busy):.1f}; idle per step: median {statistics.median(w - b for w, b in zip(wall, busy)):.1f}")
    n = len(steps[0])
    if all(len(st) == n for st in steps):
        print("per launch (median over the steps): duration us | gap before it us")
        for k in range(n):
            d = statistics.median((st[k][1] - st[k][0]) / 1e3 for st in steps)
            gaps = []
            for j, st in enumerate(steps):
                prev_end = st[k - 1][1] if k > 0 else (steps[j - 1][-1][1] if j > 0 else None)
                if prev_end is not None:
                    gaps.append((st[k][0] - prev_end) / 1e3)
            print(f"  {k:2d} {steps[0][k][2]:34s} {d:8.2f} | {statistics.median(gaps):7.2f}  (max gap {max(gaps):7.2f})")
    for st in steps[-nshow:]:
        t0 = st[0][0]
        print("step:")
        prev = None
        for s, e, nme in st:
            print(f"   +{(s - t0) / 1e3:8.2f}  {nme:34s} {(e - s) / 1e3:8.2f} us" + (f"   gap {(s - prev) / 1e3:6.2f}" if prev is not None else ""))
            prev = e


if __name__ == "__main__":
    main()
