"""tools/conv_phase_trace.py -- instruction-level account of ONE deep sparse-conv launch (isf_sparse_conv_phase_trace).

    python tools/conv_phase_trace.py [--level 3] [--batch 4] [--points 300000] [--reps 3] > profiles/r05_att_256.txt

The image's rocprofv3 has no thread-trace decoder (`rocprofv3 --att` ends with "rocprof-trace-decoder library path not
found"; no network to fetch it), so the kernel keeps its own trace: every wave stamps the shader clock (s_memtime) at the
top of each step, after its `s_waitcnt vmcnt(0)`, after the `s_barrier` and after issuing the next step's loads; the
multiply section (LDS fragment reads + MFMAs) is what is left until the next top.  From those stamps:

  * where a wave's cycles go (wait for its loads / wait at the barrier / issue the next loads / multiply), totals and
    per-step percentiles, split by how many of the wave's two 16-row groups multiply in the step;
  * cycles per MFMA as a wave sees them (multiply-phase cycles regressed on the step's MFMA count; 16 = the pipe alone);
  * per SIMD (HW_ID): for how much of the launch 0 / 1 / 2 / 3 resident waves are inside a multiply section at once, and
    the MFMA cycles the SIMD was handed against the launch span -- the matrix-pipe busy share the PMC counter reports
    chip-wide, here per SIMD with its cause;
  * what the stamps cost (traced launch against the untraced one)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HDR = 8
STEP = 8              # dwords per step record: top, after wait, after barrier, after issue, after index reads, after gathers
MFMA_CYC = 16            # v_mfma_f32_16x16x32_f16: 4 passes x 4 cycles


def pct(a, qs=(10, 50, 90)):
    import numpy as np
    return " ".join(f"p{q} {np.percentile(a, q):.0f}" for q in qs)


def analyse(wg, waves, C, nt_cols, untraced_us, plain_span_us):
    import numpy as np
    live = wg[:, 3] != 0
    wg, waves = wg[live], waves[live].astype(np.int64) & 0xffffffff
    nwg, nw, _ = waves.shape
    nch = C // 32
    loop_wall_us = (wg[:, 2] - wg[:, 1]) * 0.01
    span_us = (wg[:, 3].max() - wg[:, 0].min()) * 0.01
    print(f"launch: {nwg} workgroups x {nw} waves; traced launch span {span_us:.1f} us (per-workgroup stamps only: "
          f"{plain_span_us:.1f} us; untraced kernel in a loop: {untraced_us:.1f} us) -> the stamps cost "
          f"{100 * (span_us / plain_span_us - 1):.0f} % on top of the plain trace")
    tot = dict(wait=0.0, bar=0.0, issue=0.0, mul=0.0)
    rows = []          # per (wave, step): wait, bar, issue, mul, mfmas
    sub = []           # per (wave, step): the issue phase's index reads | gathers | weight DMA
    clk = []
    simd_iv = {}       # (xcc, se, cu, simd) -> list of (start, end, mfmas) in absolute cycles
    for b in range(nwg):
        xcc = int(wg[b, 6]) & 0xf
        taps_wg = [k for k in range(27) if (int(waves[b, 0, 6]) >> k) & 1]
        for w in range(nw):
            h = waves[b, w]
            steps = int(h[3])
            if steps == 0:
                continue
            base = (int(h[1]) << 32) | int(h[0])
            st = h[HDR:HDR + STEP * steps].reshape(steps, STEP)
            rel = (st - (base & 0xffffffff)) & 0xffffffff                      # cycles since loop entry
            end = (int(h[7]) - (base & 0xffffffff)) & 0xffffffff
            top, twait, tbar, tiss = rel[:, 0], rel[:, 1], rel[:, 2], rel[:, 3]
            nxt = np.append(top[1:], end)
            g0, g1 = int(h[4]), int(h[5])
            ntaps = len(taps_wg)
            tap_of = np.array([taps_wg[s % ntaps] for s in range(steps)])
            groups = ((g0 >> tap_of) & 1) + ((g1 >> tap_of) & 1)
            mf = groups * (nt_cols * 3)
            wait, bar, iss, mul = twait - top, tbar - twait, tiss - tbar, nxt - tiss
            tidx, tgath = rel[:, 4], rel[:, 5]
            sub.append(np.stack([(tidx - tbar) & 0xffffffff, (tgath - tidx) & 0xffffffff, (tiss - tgath) & 0xffffffff], 1))
            rows.append(np.stack([wait, bar, iss, mul, mf], 1))
            for k, v in zip(("wait", "bar", "issue", "mul"), (wait, bar, iss, mul)):
                tot[k] += float(v.sum())
            clk.append(end / max(loop_wall_us[b], 1e-9))                       # cycles per us = MHz
            hw = int(h[2])
            key = (xcc, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3)
            lst = simd_iv.setdefault(key, [])
            for s in range(steps):
                if mf[s]:
                    lst.append((base + int(tiss[s]), base + int(nxt[s]), int(mf[s])))
            lst.append((base, base + int(end), -1))                           # residency marker
    R = np.concatenate(rows)
    mhz = float(np.median(clk))
    allc = sum(tot.values())
    print(f"shader clock during the loop: {mhz:.0f} MHz (median over waves: loop cycles / loop wall time)")
    print(f"\n== where a wave's loop cycles go (sum over {len(rows)} waves, {len(R)} wave-steps)")
    for k, name in (("wait", "s_waitcnt vmcnt(0): own gathers + weight DMA of this step"), ("bar", "s_barrier: the slowest wave's loads"),
                    ("issue", "issue of the next step's loads (index reads, 4 gathers, 4 DMA pieces)"),
                    ("mul", "multiply section (16 ds_read_b128 + 24 MFMAs per active row group)")):
        print(f"   {100 * tot[k] / allc:5.1f} %  {name}")
    SB = np.concatenate(sub).astype(np.float64)
    SB = SB[(SB < 1e6).all(1)]
    if SB.size and SB.sum() > 0:
        tot_iss = SB.sum()
        print(f"   the issue phase in three pieces: index ds_reads + their wait {100 * SB[:, 0].sum() / tot_iss:.0f} % ({pct(SB[:, 0])}), "
              f"the <= 4 gathers {100 * SB[:, 1].sum() / tot_iss:.0f} % ({pct(SB[:, 1])}), the 4 weight-DMA pieces "
              f"{100 * SB[:, 2].sum() / tot_iss:.0f} % ({pct(SB[:, 2])}) -- each piece ends in a ~70-cycle stamp")
    ideal = R[:, 4].sum() * MFMA_CYC
    print(f"   MFMA issue cycles the waves own: {100 * ideal / allc:.1f} % of their loop cycles "
          f"({R[:, 4].sum() / len(rows):.0f} MFMAs per wave; a wave alone on a SIMD would need {100 * ideal / tot['mul']:.0f} % "
          f"of its multiply-section cycles)")
    print("\n== per step, cycles (p10 p50 p90), by active row groups of the wave in that step")
    for g in (0, 1, 2):
        sel = R[:, 4] == g * nt_cols * 3
        if sel.sum() < 10:
            continue
        r = R[sel]
        print(f"   {g} groups ({100 * sel.mean():4.1f} % of wave-steps): wait {pct(r[:, 0])} | barrier {pct(r[:, 1])} | issue "
              f"{pct(r[:, 2])} | multiply {pct(r[:, 3])} | whole step {pct(r[:, :4].sum(1))}")
    sel = R[:, 4] > 0
    A = np.stack([R[sel, 4], np.ones(sel.sum())], 1)
    coef = np.linalg.lstsq(A, R[sel, 3], rcond=None)[0]
    print(f"   multiply cycles = {coef[0]:.1f} x MFMAs + {coef[1]:.0f}  (16.0 x = the pipe to itself); a step of a wave: "
          f"mean {R[:, :4].sum(1).mean():.0f} cycles = {R[:, :4].sum(1).mean() / mhz:.2f} us")
    # ---- per SIMD
    print(f"\n== per SIMD ({len(simd_iv)} SIMDs seen): resident waves inside a multiply section at the same time")
    occ_hist = np.zeros(6)
    res_hist = np.zeros(6)
    util, util_res = [], []
    span_cyc = span_us * mhz      # the launch span in shader cycles (the counters of different XCDs are not comparable)
    for key, lst in simd_iv.items():
        ev = []
        mf_cycles = 0
        for s, e, m in lst:
            if m >= 0:
                ev.append((s, 0, 1)); ev.append((e, 0, -1)); mf_cycles += m * MFMA_CYC
            else:
                ev.append((s, 1, 1)); ev.append((e, 1, -1))
        ev.sort()
        cur = [0, 0]
        last = ev[0][0]
        first_res, last_res = None, None
        for t, kind, d in ev:
            if cur[1] > 0:
                occ_hist[min(cur[0], 5)] += t - last
                res_hist[min(cur[1], 5)] += t - last
            last = t
            cur[kind] += d
        res = [(s, e) for s, e, m in lst if m < 0]
        lo, hi = min(s for s, e in res), max(e for s, e in res)
        util.append(mf_cycles / span_cyc)
        util_res.append(mf_cycles / (hi - lo))
    occ = occ_hist / occ_hist.sum()
    resd = res_hist / res_hist.sum()
    print("   while at least one wave is in its loop on the SIMD, waves inside a multiply section: " +
          ", ".join(f"{i}: {100 * occ[i]:.1f} %" for i in range(5)))
    print("   waves resident in their loop on the SIMD over the same time: " + ", ".join(f"{i}: {100 * resd[i]:.1f} %" for i in range(1, 5)))
    util, util_res = np.array(util), np.array(util_res)
    print(f"   MFMA cycles handed to a SIMD / launch span: mean {100 * util.mean():.1f} % p10 {100 * np.percentile(util, 10):.1f} "
          f"p90 {100 * np.percentile(util, 90):.1f} max {100 * util.max():.1f} %   (the PMC's SQ_VALU_MFMA_BUSY_CYCLES share, "
          f"per SIMD)")
    print(f"   MFMA cycles / the SIMD's own busy span (first loop entry .. last loop exit): mean {100 * util_res.mean():.1f} % "
          f"p10 {100 * np.percentile(util_res, 10):.1f} p90 {100 * np.percentile(util_res, 90):.1f} max {100 * util_res.max():.1f} %")
    busiest = np.argsort(util)[-max(1, len(util) // 10):]
    print(f"   the busiest tenth of the SIMDs: MFMA share of the launch span {100 * util[busiest].mean():.1f} %, of their own "
          f"span {100 * util_res[busiest].mean():.1f} %")
    return dict(mhz=mhz, shares={k: tot[k] / allc for k in tot}, mfma_share=ideal / allc, cyc_per_mfma=float(coef[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=3, choices=[2, 3])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dump", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from conv_trace import level_rulebooks
    from isfusion_amd import spconv
    dev = torch.device("cuda", 0)
    pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, args.batch, args.points, 0)]
    rb = level_rulebooks(pts, args.batch, args.level)
    C = 256 if args.level == 3 else 128
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(rb.num_in, C, generator=g).to(dev)
    w = (torch.randn(3, 3, 3, C, C, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dev)
    packed = spconv.pack_filters_f16x3(w)
    xs = spconv.to_split(x)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    order = spconv.tile_order(rb, C, C)
    pairs = int((rb.nbr.view(27, rb.stride)[:, :rb.num_out] >= 0).sum().item())
    print(f"# phase trace of spconv_f16x3_kernel, level {args.level}: {rb.num_out} rows, {C} -> {C}, {pairs} pairs "
          f"({pairs / rb.num_out:.1f} per row), B = {args.batch} x {args.points} points, production tile order")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib_call = lambda: spconv.sparse_conv_trace(xs, packed, 27, C, C, rb, scale, shift, None, True, order=order)
    for _ in range(3):
        lib_call()
    torch.cuda.synchronize()
    spans = []
    for _ in range(args.reps):
        _, tr = lib_call()
        torch.cuda.synchronize()
        t = tr.cpu().numpy()
        t = t[t[:, 3] != 0]
        spans.append((t[:, 3].max() - t[:, 0].min()) * 0.01)
    plain_span = float(np.median(spans))
    # untraced: the same layer through the production entry, 20 launches between two events
    from isfusion_amd import _lib
    lib = _lib.load()
    out = torch.empty(rb.num_out * C * 4, dtype=torch.uint8, device=dev)

    def prod():
        _lib.check(lib.isf_sparse_conv_forward_f16x3_ordered(
            _lib.ptr(xs), rb.num_in, C, _lib.ptr(packed), 27, C, _lib.ptr(rb.nbr), rb.stride, rb.num_out, _lib.ptr(scale),
            _lib.ptr(shift), None, 1, _lib.ptr(out), 0, _lib.ptr(order), _lib.stream()), "conv")
    for _ in range(5):
        prod()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        prod()
    e1.record()
    torch.cuda.synchronize()
    untraced = e0.elapsed_time(e1) * 1000 / 20
    for r in range(args.reps):
        ys, wgt, wv = spconv.sparse_conv_phase_trace(xs, packed, 27, C, C, rb, scale, shift, None, True, order=order)
        torch.cuda.synchronize()
    same = bool(torch.equal(ys, out))
    print(f"# traced launch output == production output bit for bit: {same}")
    res = analyse(wgt.cpu().numpy(), wv.cpu().numpy(), C, 8, untraced, plain_span)
    flops = 2.0 * pairs * C * C
    print(f"\nalgorithmic {flops / 1e9:.2f} GFLOP per launch; untraced {untraced:.1f} us = {flops / untraced / 1e6:.0f} TFLOP/s = "
          f"{flops / untraced / 1e6 / 833.3:.3f} of the f16x3 roofline (2500 / 3 TFLOP/s)")
    if args.dump:
        np.savez_compressed(args.dump, wg=wgt.cpu().numpy(), waves=wv.cpu().numpy())
    return res


if __name__ == "__main__":
    main()
