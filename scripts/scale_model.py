"""Predicted time of ONE sharded N x N evaluation on 2 / 4 / 8 GPUs from measured single-GPU step times and a link model, so
that the first SCALE record falsifies a number instead of an adjective (the schedule itself: csrc/gphip_multi.inc
group_eval_run; no multi-GPU box has run it yet).

Input: gpurun_out/owner_path_<N>.json (scripts/gpu_owner_path.py: per outer panel, with the chip to itself, the time of the
panel factorisation, of the look-ahead update LA(k) of panel k+1 and of the trailing update REST(k) of everything behind it),
and the link parameters alpha (seconds per collective call, launch to completion of an empty message) and beta (bytes / s a
receiver sees from one broadcast).

Model = the dependency graph of group_eval_run replayed with those durations, three streams per rank:
    panel stream   LA(k) on owner(k+1): waits for panel k's arrival and for REST(k-1)'s piece on panel k+1;  then factor(k+1)
    comm stream    broadcast of panel k+1: per tile column (bcast_chunks) as the columns become final, or one message;
                   waits for the receive buffer (REST(k-2) done)
    main stream    REST(k) on every rank: its share of the trailing panels (by tile count), owner(k+2) does panel k+2 first
A rank's REST share shrinks with the world, but a launch never runs faster than `floor_us` (tail of a grid that no longer
fills 256 CUs).  Panel-stream work and REST share one GPU: the makespan is at least each rank's total work.
Host term (round 6): nothing of step k starts before the host thread has issued it -- `issue_us` per rank and outer panel
(measured: option "last_issue_us" of an 8-virtual-rank evaluation / 8 ranks / panels, scripts/gpu_multi_overhead.py), times
the number of local ranks when ONE thread drives all of them (a one-process multi-device handle).
Speed-ups are printed against the model's own one-rank sharded time AND against the plain one-GPU evaluation (`--plain-ms`,
what bench.py's `strong_speedup` divides by).
   python scripts/scale_model.py owner_path_32768.json [more.json ..] [--links a_us:b_GBs,..] [--plain-ms 177.1] [--issue-us 35]"""
import json
import sys


def simulate(data, mode, W, alpha_us, beta_gbs, chunks=True, two_hop=False, split_rest=True, floor_us=25.0, launch_us=6.0, issue_us=0.0,
             one_thread=False, final_at_end=False, first_ready=None, owner_yield=False):
    N, P = data["N"], data["panel_tiles"]
    m = data["modes"][mode]
    f, la, rest = m["factor_us"], m["la_us"], m["rest_us"]
    nouter = len(f)
    Nt = (N + 127) // 128
    R = Nt + 1
    width = [min(P, Nt - j * P) for j in range(nouter)]
    # tiles of outer panel k (what a broadcast moves), tile columns c of the panel hold R - (kP + c) tiles
    col_tiles = [[R - (k * P + c) for c in range(width[k])] for k in range(nouter)]
    tile_bytes = 128 * 128 * 8
    # weight of outer panel j in a trailing update (tiles touched): rows below its top x its width
    wj = [(R - j * P) * width[j] for j in range(nouter)]
    beta = beta_gbs * 1e9 * (W / 2.0 if two_hop and W > 2 else 1.0)          # two-hop: scatter + all-gather over all links
    alpha = alpha_us * (2.0 if two_hop and W > 2 else 1.0)

    def own(j):
        return j % W

    def issued(k):                                         # host time at which step k's operations exist on the streams
        return (k + 1) * issue_us * (W if one_thread else 1)

    p = [0.0] * W
    c = [0.0] * W
    mn = [0.0] * W
    work = [0.0] * W
    B = [[0.0] * W for _ in range(nouter)]                 # arrival of panel k on rank i
    first_done = [[0.0] * W for _ in range(nouter)]
    rest_done = [[0.0] * W for _ in range(nouter)]

    def factor_and_broadcast(k):
        o = own(k)
        p[o] = max(p[o], issued(k - 1))
        start = p[o]
        p[o] += f[k]
        work[o] += f[k]
        F = p[o]
        if W == 1:
            B[k][0] = F
            return
        nin = width[k]
        for i in range(W):
            t = max(c[i], issued(k - 1))
            if k >= 3:
                t = max(t, rest_done[k - 3][i])
            if chunks:
                for s2 in range(nin):
                    # (a dataflow panel launch -- dist_panel_df -- has no 'tile column final' events: every column is ready when it ends)
                    # dist_panel_df = 3 counts finished tiles per column inside the launch and the broadcast stream waits on the
                    # counter: column 0 is final `first_ready` of the way through the launch (measured, profiles/r06_colsig_overlap.txt:
                    # the look-ahead update of the whole panel comes first), the last one at its end
                    if first_ready is not None:
                        ready = start + (first_ready + (1.0 - first_ready) * s2 / max(nin - 1, 1)) * (F - start)
                    else:
                        ready = F if final_at_end else start + (s2 + 1) * (F - start) / nin
                    t = max(t, ready) + alpha + col_tiles[k][s2] * tile_bytes / beta * 1e6
            else:
                t = max(t, F) + alpha + sum(col_tiles[k]) * tile_bytes / beta * 1e6
            c[i] = t
            B[k][i] = F if i == o else t            # (the owner reads its own storage: ready when factored)

    factor_and_broadcast(0)
    deferred = [0.0] * W                                   # owner_yield: remainder of REST(k-1) still to run on that rank (duration)
    for k in range(nouter):
        if k + 1 < nouter:
            o = own(k + 1)
            t = max(p[o], B[k][o], issued(k))
            if k >= 1:
                t = max(t, first_done[k - 1][o] if split_rest else rest_done[k - 1][o])
            p[o] = t + la[k]
            work[o] += la[k]
            factor_and_broadcast(k + 1)
        lo = k + 2 if k + 1 < nouter else k + 1
        tot = sum(wj[j] for j in range(lo, nouter)) or 1
        for i in range(W):
            if owner_yield and W > 1 and k + 1 < nouter and own(k + 1) == i:
                # option dist_owner_yield: this rank's trailing work stands back until its panel launch factor(k+1) has ended,
                # then the deferred remainder of REST(k-1) runs, then REST(k)
                mn[i] = max(mn[i], p[i]) + deferred[i]
                if k >= 1:
                    rest_done[k - 1][i] = mn[i]
                deferred[i] = 0.0
            t = max(mn[i], B[k][i], issued(k))
            mine = [j for j in range(lo, nouter) if own(j) == i]
            share = sum(wj[j] for j in mine) / tot
            dur_all = 0.0
            if mine or (i == 0):                           # (rank 0 also updates the corner / rhs tile)
                dur_all = max(rest[k] * share, floor_us if mine else 0.0) + launch_us
            if W > 1 and split_rest and k + 2 < nouter and own(k + 2) == i:
                d1 = max(rest[k] * wj[k + 2] / tot, floor_us) + launch_us
                t += d1
                first_done[k][i] = t
                rem = max(dur_all - d1, 0.0) + (launch_us if len(mine) > 1 else 0.0)
                if owner_yield:
                    deferred[i] = rem                      # goes out behind factor(k + 2), next step
                else:
                    t += rem
                work[i] += max(dur_all, d1)
            else:
                t += dur_all
                first_done[k][i] = t
                work[i] += dur_all
            rest_done[k][i] = t
            mn[i] = t
    path = max(max(mn), max(p)) + (alpha_us if W > 1 else 0.0)
    return max(path, max(work)), path, max(work)


def main():
    args = sys.argv[1:]
    paths, links, plain_ms, issue_us = [], None, 177.07, 0.0
    i = 0
    while i < len(args):
        if args[i] == "--links":
            links = [tuple(float(x) for x in it.split(":")) for it in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--plain-ms":
            plain_ms = float(args[i + 1])
            i += 2
        elif args[i] == "--issue-us":
            issue_us = float(args[i + 1])
            i += 2
        else:
            paths.append(args[i])
            i += 1
    paths = paths or ["gpurun_out/owner_path_32768.json"]
    links = links or [(10.0, 120.0), (20.0, 60.0), (40.0, 30.0)]
    # (name, W -> (step-time set, simulate() options)).  The library's default since round 6 (gphip_dist_begin, gphip_multi.inc
    # group_eval_run): dataflow panels whose tile columns are handed over by counters (dist_panel_df = 3); from 4 ranks every
    # message as scatter + in-place all-gather over all links.
    df0 = lambda two_hop: ("df0_fuse1", dict(chunks=True, two_hop=two_hop))
    df2 = lambda two_hop: ("df2_fuse0", dict(chunks=True, final_at_end=True, two_hop=two_hop))
    df3 = lambda two_hop, yld=False: ("df2_fuse0", dict(chunks=True, first_ready=0.36, two_hop=two_hop, owner_yield=yld))
    variants = [("DEFAULT round 6: dataflow panels with column signals (dist_panel_df=3); from 4 ranks two-hop + the owner's trailing work yields to its panel launch", lambda W: df3(W >= 4, W >= 4)),
                ("the same without the owner yielding (the model lets a panel launch and a trailing update of one GPU overlap at no cost: optimistic)", lambda W: df3(W >= 4)),
                ("fallback default (no stream-ordered wait on the device): dataflow panels at 2 ranks; per-tile-column panels + two-hop from 4 ranks", lambda W: df2(False) if W == 2 else df0(W >= 4)),
                ("round-5 default: per-tile-column panels, plain ncclBroadcast", lambda W: df0(False)),
                ("dataflow panels (dist_panel_df=2), plain broadcast", lambda W: df2(False)),
                ("dataflow panels + two-hop from 4 ranks", lambda W: df2(W >= 4)),
                ("dataflow panels with column signals (dist_panel_df=3), plain broadcast", lambda W: df3(False))]
    for path in paths:
        data = json.load(open(path))
        print(f"# N = {data['N']}, outer panel = {data['panel_tiles']} tiles ({path}); times in ms; host issue {issue_us:.0f} us per rank and panel; "
              f"plain one-GPU evaluation {plain_ms:.1f} ms")
        for alpha, beta in links:
            print(f"## alpha = {alpha:.0f} us per collective, beta = {beta:.0f} GB/s per receiver")
            print("| schedule | 1 GPU | 2 GPUs | 4 GPUs | 8 GPUs | at 8: vs own 1-rank time | at 8: vs plain one-GPU | one host thread for all 8 ranks | bound at 8 |")
            print("|---|---|---|---|---|---|---|---|---|")
            for name, pick in variants:
                t = {}
                why = ""
                for W in (1, 2, 4, 8):
                    mode, kw = pick(W)
                    tot, cp, wk = simulate(data, mode, W, alpha, beta, issue_us=issue_us, **kw)
                    t[W] = tot
                    if W == 8:
                        why = "owner chain + links" if cp >= wk else "work per rank"
                mode, kw = pick(8)
                one, _, _ = simulate(data, mode, 8, alpha, beta, issue_us=issue_us, one_thread=True, **kw)
                print(f"| {name} | {t[1] / 1e3:.1f} | {t[2] / 1e3:.1f} | {t[4] / 1e3:.1f} | {t[8] / 1e3:.1f} | {t[1] / t[8]:.2f}x | "
                      f"{plain_ms * 1e3 / t[8]:.2f}x | {one / 1e3:.1f} ms = {plain_ms * 1e3 / one:.2f}x | {why} |")
        for label, mode in (("per-tile-column panels", "df0_fuse1"), ("dataflow panels", "df2_fuse0")):
            if mode in data["modes"]:
                m = data["modes"][mode]
                print(f"   owner chain, {label}: factor {sum(m['factor_us']) / 1e3:.1f} ms + look-ahead {sum(m['la_us']) / 1e3:.1f} ms; "
                      f"trailing work {sum(m['rest_us']) / 1e3:.1f} ms / 8 = {sum(m['rest_us']) / 8e3:.1f} ms per rank; "
                      f"factor bytes per receiver {data['N'] ** 2 * 4 / 1e9:.2f} GB "
                      f"(= {data['N'] ** 2 * 4 / 1e9 / 0.06:.0f} ms over ONE 60 GB/s link, {data['N'] ** 2 * 4 / 1e9 / 0.24:.0f} ms two-hop at 8 ranks)")
        print("   term that decides 6x (= %.1f ms at 8 GPUs): the owner chain (factor + look-ahead, 33.6 ms per-tile-column / 23.5 ms dataflow) is SERIAL across\n"
              "   panels and each hop of it also waits for the last tile column's transfer (per-tile-column panels, dataflow panels with column signals)\n"
              "   or the whole panel's (dataflow panels without signals); a rank's trailing share (21 ms) hides under it, not the other way round." % (plain_ms / 6.0))

if __name__ == "__main__":
    main()
