"""tools/stream_split.py -- does splitting the batch of 4 over two HIP streams pay?  (prologue / epilogue phases of one
half overlap the MFMA loops of the other; single-round launches of the deep levels fill the idle SIMDs of each other.)
One process: (a) lb(4 frames) on one stream, (b) two host threads, each lb(2 frames) on its own stream.
Prints frames/s of both.   python tools/stream_split.py [--steps 40]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--points", type=int, default=300000)
    args = ap.parse_args()
    import torch
    import isfusion_amd as m
    import bench
    dev = torch.device("cuda", 0)
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    sets = [[torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, 4, args.points, fs)] for fs in range(2)]
    for i in range(5):
        lb(sets[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        lb(sets[s % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"one stream, batch 4:  {4 * args.steps / dt:8.1f} frames/s  {dt / args.steps * 1e3:.3f} ms/step")

    def worker(half, stream, n):
        with torch.cuda.stream(stream):
            for s in range(n):
                lb(sets[s % 2][2 * half:2 * half + 2])
            stream.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for n in (5, args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(h, streams[h], n)) for h in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"two streams, 2 x batch 2: {4 * args.steps / dt:8.1f} frames/s  {dt / args.steps * 1e3:.3f} ms per 4 frames")

    # (c) two forwards of the FULL batch in flight (consecutive batches pipelined on two streams)
    def worker_full(stream, n, off):
        with torch.cuda.stream(stream):
            for s in range(n):
                lb(sets[(s + off) % 2])
            stream.synchronize()

    for n in (5, args.steps // 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker_full, args=(streams[h], n, h)) for h in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"two streams, 2 x batch 4 in flight: {4 * 2 * (args.steps // 2) / dt:8.1f} frames/s  "
          f"{dt / (2 * (args.steps // 2)) * 1e3:.3f} ms per batch of 4")


if __name__ == "__main__":
    main()
