"""Python model of the control flow of the compacted-rows dW kernel (isf_spconv_bwd.hip under ISF_WGRAD_COMPACT: 256-row
batches, ballot ranks, the LDS list, the < 4 pairs carried over, the padded final group): every live (input row, output
row) pair of a chunk is fed to the MFMAs exactly once, in row order, and only the final group is padded.  CPU only; the
kernel itself has not run on hardware yet.

    python tools/probes/wgrad_compact_model.py"""
import numpy as np


def emulate(nk, r_begin, r_end, batch=256):
    lst, out = [None] * (batch + 4), []

    def fetch(r0):
        return [[(nk[r0 + q * 64 + l] if r0 + q * 64 + l < r_end else -1) for l in range(64)] for q in range(4)]

    ahead, carried, r0 = fetch(r_begin), 0, r_begin
    while r0 < r_end:
        cur, ahead, n = ahead, fetch(r0 + batch), carried
        for q in range(4):
            live = [v >= 0 for v in cur[q]]
            rank = np.cumsum([0] + live[:-1])                      # mbcnt of the ballot
            for l in range(64):
                if live[l]:
                    lst[n + rank[l]] = (cur[q][l], r0 + q * 64 + l)
            n += sum(live)
        last = r0 + batch >= r_end
        groups = (n + 3) >> 2 if last else n >> 2
        if last:
            for l in range(max(0, 4 * groups - n)):
                lst[n + l] = (-1, 0)
        out += [lst[4 * g + ks] for g in range(groups) for ks in range(4)]
        carried = n - 4 * groups
        if not last:
            keep = [lst[4 * groups + l] for l in range(carried)]
            lst[:carried] = keep
        r0 += batch
    return out


def main():
    rng = np.random.default_rng(0)
    for trial in range(300):
        n_out = int(rng.integers(1, 3000))
        nk = np.where(rng.random(n_out) < rng.random(), rng.integers(0, 10 ** 6, n_out), -1)
        rb = int(rng.integers(0, n_out))
        re_ = int(rng.integers(rb, n_out + 1))
        if rng.random() < 0.3:
            rb, re_ = 0, n_out
        got = emulate(list(nk), rb, re_)
        want = [(int(nk[r]), r) for r in range(rb, re_) if nk[r] >= 0]
        assert [e for e in got if e[0] >= 0] == want, trial
        assert len(got) % 4 == 0 and all(e == (-1, 0) for e in got[len(want):]), trial
    print("300 random chunks: every live pair once, in order; only the tail padded")


if __name__ == "__main__":
    main()
