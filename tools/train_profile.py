"""Steady-state kernel time of one training step: per-kernel totals of two rocprofv3 traces of tools/train_step.py that
differ only in the number of steps (the first step carries MIOpen / hipBLASLt find runs -- naive_conv_* etc. -- that would
swamp a single trace).  usage: python tools/train_profile.py SHORT_DB LONG_DB steps_short steps_long"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocpd_summary import kernel_stats  # noqa: E402


def main():
    a, b = kernel_stats(sys.argv[1]), kernel_stats(sys.argv[2])
    n = int(sys.argv[4]) - int(sys.argv[3])
    rows = []
    for k, s in b.items():
        d = s["tot"] - a.get(k, {"tot": 0})["tot"]
        c = s["n"] - a.get(k, {"n": 0})["n"]
        if c > 0:
            rows.append((d / n / 1e3, c / n, k))
    rows.sort(reverse=True)
    print(f"# per training step (difference of the two traces / {n} steps): {sum(r[0] for r in rows) / 1e3:.2f} ms of kernels, "
          f"{sum(r[1] for r in rows):.0f} launches")
    print(f"{'us/step':>10s} {'calls':>7s}  kernel")
    for us, c, k in rows[:70]:
        print(f"{us:10.1f} {c:7.1f}  {k}")


if __name__ == "__main__":
    main()
