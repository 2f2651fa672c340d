#!/usr/bin/env python3
"""Developer tool (GPU box, library built with -DCPF_STREAM_TIMELINE): when does each persistent wave of the streaming
step kernel start and end, how many tiles and rounds did it do.  python tools/stream_timeline.py [--opt k=v ...]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--label", default="")
    ap.add_argument("--pre-steps", type=int, default=5, help="steps after the sort and before the measured launch")
    ap.add_argument("--variant", type=int, default=4)
    ap.add_argument("--D", type=float, default=0.0, help="diffusion coefficient of the measured launches")
    ap.add_argument("--groups", action="store_true", help="also per-group (= per-CU) and per-XCD end times")
    ap.add_argument("--mesh3d", action="store_true", help="the 245 760-cell 3-D mesh (tools/_cases.py: box3d), swirl field")
    ap.add_argument("--tjunction", action="store_true", help="the reference's TJunction tutorial mesh (248 000 cells), split flow u0 = 3")
    ap.add_argument("--case", default=None, help="tools/_cases.py: pitz | box3d | tjunction | octagons | pentagons | dodecagons | hexgrid")
    ap.add_argument("--census", action="store_true", help="busy lanes per round index, sit-outs, rounds against the largest visit count of a tile")
    args = ap.parse_args()
    import torch
    from cudaparticlesfoam_amd.api import Context
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = int(args.particles)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _cases import POLY_CASES, POLY_DT, make_case
    case = args.case or ("tjunction" if args.tjunction else ("box3d" if args.mesh3d else "pitz"))
    mesh, x, y, z, c, fields = make_case(case, ctx, torch, n, dev, "swirl" if case == "box3d" else None)
    dt = POLY_DT if case in POLY_CASES else 1e-4
    g = torch.arange(n, dtype=torch.int64, device=dev)
    ctx.set_option("stats", 0); ctx.set_option("step_variant", args.variant)
    for kv in args.opt:
        k, v = kv.split("="); ctx.set_option(k, float(v))
    ctx.sort_by_cell_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), c.data_ptr(), g.data_ptr(), n)
    NCEN = 16 * 65 + 16 + 256 + 16
    tl = torch.zeros(8 * 16384 + NCEN, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    for _ in range(args.pre_steps):
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g) if args.D > 0 else None, None, n, dt, args.D, 0, 1, 0)
    torch.cuda.synchronize()
    tl.zero_()
    if args.census:
        ctx.set_option("stream_debug", 4)        # the census' global atomics distort the timeline: only on request
    ctx.timing_enable(True)
    ctx.step_dev(p(x), p(y), p(z), p(c), p(g) if args.D > 0 else None, p(tl), n, dt, args.D, 5, 1, 0)
    launches, ms = ctx.timing_read()
    raw_all = tl.cpu().numpy()
    raw, cen = raw_all[:8 * 16384], raw_all[8 * 16384:].astype(np.float64)
    stride = 8 if args.variant == 4 else 4          # variant 4 builds also report waits: see below
    t = raw[: (raw.size // stride) * stride].reshape(-1, stride)
    idx = np.nonzero(t[:, 1] > 0)[0]
    t = t[idx]
    waits = t[:, 4:7] if stride == 8 else None      # ticks waited for missing records | at tile ends | rounds with a miss
    t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
    base = t0.min()
    start = (t0 - base) * 10.0 / 1e3          # us (100 MHz ticks)
    end = (t1 - base) * 10.0 / 1e3
    q = lambda a: [round(float(v), 2) for v in np.percentile(a, [0, 5, 25, 50, 75, 95, 100])]   # noqa: E731
    out = dict(label=args.label, opts=args.opt, kernel_ms=round(ms / launches, 4), waves=int(t.shape[0]),
               start_us_pct=q(start), end_us_pct=q(end), life_us_pct=q(end - start),
               tiles_pct=q(t[:, 2]), rounds_per_tile=round(float(t[:, 3].sum() / max(1, t[:, 2].sum())), 3),
               us_per_tile_pct=q((end - start) / np.maximum(t[:, 2], 1)),
               slot_busy_fraction=round(float((end - start).sum() / (t.shape[0] * end.max())), 3))
    if waits is not None and waits.any():
        life = float((end - start).sum())
        wrec = float(waits[:, 0].sum()) * 10.0 / 1e3
        wend = float(waits[:, 1].sum()) * 10.0 / 1e3
        miss_rounds = waits[:, 2].astype(np.float64)
        out["wait_for_records_fraction_of_wave_time"] = round(wrec / life, 3)
        out["wait_at_tile_end_fraction_of_wave_time"] = round(wend / life, 3)
        out["miss_rounds_per_tile"] = round(float(miss_rounds.sum() / max(1, t[:, 2].sum())), 3)
        out["us_per_record_wait"] = round(wrec / max(1.0, float(miss_rounds.sum())), 3)
        out["us_per_tile_end_wait"] = round(wend / max(1.0, float(t[:, 2].sum())), 3)
    if args.census and cen.any():
        H = cen[:16 * 65].reshape(16, 65)
        sat = cen[16 * 65:16 * 65 + 16]
        J = cen[16 * 65 + 16:16 * 65 + 16 + 256].reshape(16, 16)        # [rounds of the tile][largest visit count of a lane]
        jobs = cen[16 * 65 + 16 + 256:]
        tiles = J.sum()
        rounds_r = H.sum(1)                                             # tiles that ran a round with index r
        busy_r = (H * np.arange(65)[None, :]).sum(1)                    # busy lanes entering round r
        out["census"] = dict(
            tiles=int(tiles),
            rounds_per_tile=round(float(rounds_r.sum() / tiles), 3),
            ideal_rounds_per_tile=round(float((J.sum(0) * np.arange(16)).sum() / tiles), 3),
            tiles_running_round=[round(float(v / tiles), 4) for v in rounds_r[:10]],
            busy_lanes_per_tile_entering_round=[round(float(v / tiles), 3) for v in busy_r[:10]],
            sitout_lanes_per_tile_in_round=[round(float(v / tiles), 3) for v in sat[:10]],
            record_requests_per_tile_in_round=[round(float(v / tiles), 3) for v in jobs[:10]],
            rounds_hist=[round(float(v / tiles), 4) for v in J.sum(1)[:12]],
            ideal_hist=[round(float(v / tiles), 4) for v in J.sum(0)[:12]],
            # had the tile stopped after R rounds: lanes still busy (to be parked), per tile
            parked_lanes_per_tile_if_capped_at=[round(float(busy_r[R] / tiles), 3) for R in range(1, 7)],
            rounds_per_tile_if_capped_at=[round(float(rounds_r[:R].sum() / tiles), 3) for R in range(1, 7)],
            # rounds with <= 8 / 16 busy lanes entering, per tile
            rounds_with_le8_busy=round(float(H[:, 1:9].sum() / tiles), 3), rounds_with_le16_busy=round(float(H[:, 1:17].sum() / tiles), 3),
            rounds_with_le8_busy_by_round=[round(float(v / tiles), 3) for v in H[:10, 1:9].sum(1)],
        )
    if args.groups:
        # per chunk-counter group (block id mod 256 = one CU: blocks go round-robin over 8 XCDs x 32 CUs) and per XCD
        grp = idx % 256
        gend = np.array([end[grp == k].max() for k in range(256)])
        gmin = np.array([end[grp == k].min() for k in range(256)])
        gtiles = np.array([t[grp == k, 2].sum() for k in range(256)])
        grounds = np.array([t[grp == k, 3].sum() for k in range(256)])
        gus = np.array([((end - start)[grp == k]).sum() for k in range(256)]) / np.maximum(gtiles, 1)
        out["group_last_end_pct"] = q(gend)
        out["group_first_end_pct"] = q(gmin)
        out["group_tiles_pct"] = q(gtiles)
        out["group_rounds_per_tile_pct"] = q(grounds / np.maximum(gtiles, 1))
        out["group_wave_us_per_tile_pct"] = q(gus)
        out["corr_end_vs_rounds"] = round(float(np.corrcoef(gend, grounds)[0, 1]), 3)
        out["corr_end_vs_us_per_round"] = round(float(np.corrcoef(gend, gus * gtiles / np.maximum(grounds, 1))[0, 1]), 3)
        out["xcd_last_end"] = [round(float(gend[np.arange(256) % 8 == k].max()), 1) for k in range(8)]
        out["xcd_mean_end"] = [round(float(gend[np.arange(256) % 8 == k].mean()), 1) for k in range(8)]
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
