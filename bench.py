"""Benchmark of the diffusion sweep (the hot path of diffusion_stereo) on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step is one diffusion iteration (red half-sweep + black half-sweep over the whole
image) of the synthetic 3840x2160 pair with 256 disparity labels, f32 potentials,
cost volume resident in HBM.  For N > 1 the image is split into N row bands, one rank
per GPU, with one halo row exchanged per neighbour after each colour (RCCL
send/recv): total work is fixed, so scaling is strong.

Prints ONE JSON line on rank 0: metric = valid (pixel, label) pairs x iterations per
second for the whole job, plus `roofline` (the dominant kernel against the 8 TB/s HBM peak,
from HIP events around every launch in the timed region; for N > 1 the slowest rank's, with
every rank's account in `per_rank`) and, at N = 1, `cpu_baseline` (the faithful
single-threaded CPU restatement of the reference rule, timed on this host on a bounded crop
of the same workload), `sustained` (the same handle over a few seconds: a kernel bound by
the vector ALU follows the clock the GPU holds) and `f64` (the same workload in the
reference's own precision).
"""
import argparse
import ctypes as C
import json
import os

# ROCr reads this when the runtime starts (the first HIP call of the process): set before torch
# can make one.  The host driver only supports dmabuf IPC; RCCL's buffer sharing needs it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from stereo_vision_amd import bands, diffstereo, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
WORKLOADS = {
    "4k_256": dict(H=2160, W=3840, D=256, dtype="f32"),
    "4k_256_f64": dict(H=2160, W=3840, D=256, dtype="f64"),  # the reference's own precision: 102 GB of phi
    "1080p_128": dict(H=1080, W=1920, D=128, dtype="f32"),
    "1080p_40": dict(H=1080, W=1920, D=40, dtype="f32"),     # the reference's own disparity count
    "1080p_40_f64": dict(H=1080, W=1920, D=40, dtype="f64"),  # ... in the reference's own precision
    "8k_256_f64": dict(H=4320, W=7680, D=256, dtype="f64"),
    "8k_256_f32": dict(H=4320, W=7680, D=256, dtype="f32"),  # 204 GB of phi: one GPU holds it
}


def valid_pairs(H, W, D):
    """(pixel, label) pairs with d <= j (the ones the reference touches)."""
    return H * sum(min(j, D - 1) + 1 for j in range(W))


def pmc_traffic(workload, kernel):
    """HBM bytes per sweep launch from the committed rocprofv3 --pmc passes
    (profiles/pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs,
    FETCH_SIZE doubled as gfx950 requires); None if no pass matches this kernel."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
    except (OSError, ValueError):
        return None
    for key, rec in table.items():
        if key.startswith(workload) and rec.get("kernel") == kernel:
            return float(rec["hbm_bytes_per_launch"])
    return None


def cpu_baseline(left, right, D, lam, budget_s=12.0):
    """Faithful O(D^2) single-thread restatement (reference rule, raster order) on a crop."""
    from oracle import pyoracle
    H, W = 4, min(left.shape[1], 384)
    crop_l, crop_r = left[:H, :W], right[:H, :W]
    g = pyoracle.OracleGraph(crop_l, crop_r, D, lam)
    iters, t0 = 0, time.perf_counter()
    while True:
        g.diffusion_act()
        iters += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or iters >= 50:
            break
    pl = valid_pairs(H, W, D)
    out = {"value": pl * iters / dt / 1e9, "unit": "G pixel*label*iters/s", "cores": 1, "kind": "port",
           "sample": "C restatement of diffusion_act (raster order, O(D^2), f64), %d iteration(s) on the "
                     "%dx%d top-left crop of the same synthetic pair, D=%d, %.1f s" % (iters, W, H, D, dt)}
    # the same restatement at the reference's own working scale (main.rs:38-47: images of about
    # 400x400, 40 disparities), where its O(D^2) cost per pixel is 40x smaller than at D = 256
    rs = min(left.shape[0], 200), min(left.shape[1], 400)   # half its rows: the sample stays a few seconds
    g = pyoracle.OracleGraph(left[:rs[0], :rs[1]], right[:rs[0], :rs[1]], 40, lam)
    t0 = time.perf_counter()
    g.diffusion_act()
    dt_rs = time.perf_counter() - t0
    out["reference_scale"] = {"value": valid_pairs(rs[0], rs[1], 40) / dt_rs / 1e9, "cores": 1,
                              "sample": "the same restatement, 1 iteration on the %dx%d top-left crop, D=40, %.1f s"
                                        % (rs[1], rs[0], dt_rs)}
    # the O(D) OpenMP port of the checkerboard sweep, all host threads OpenMP gives us
    Hf = min(left.shape[0], 96)
    f = pyoracle.FastOracle(left[:Hf], right[:Hf], D, lam, dtype="f32")
    f.iterate(1)
    t0 = time.perf_counter()
    f.iterate(2)
    dt = time.perf_counter() - t0
    out["fast_port"] = {"value": valid_pairs(Hf, left.shape[1], D) * 2 / dt / 1e9,
                        "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                        "sample": "O(D) OpenMP checkerboard port, f32, 2 iterations on the top %d rows" % Hf}
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (for libraries that print from C)."""

    def __enter__(self):
        sys.stdout.flush()
        self.keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.keep, 1)
        os.close(self.keep)
        return False


class SingleGpu:
    """One GPU: the whole image through ds_iterate, or (--band-of) one band's kernels in the
    order the band protocol launches them, without a neighbour to exchange with."""

    def __init__(self, graph):
        self.g = graph
        self.whole = graph.row_begin == 0 and graph.row_end == graph.H

    def iterate(self, n):
        g = self.g
        if self.whole:
            g.iterate(n)
            return
        fb, fe = g.fused_rows()
        if n <= 0 or fe - fb < 4:
            for _ in range(n):
                g.sweep(0)
                g.sweep(1)
            g.synchronize()
            return
        for _ in range(n):
            if fb > g.row_begin:
                g.sweep(0, g.row_begin, g.row_begin + 2)
                g.sweep_next(g.row_begin, g.row_begin + 1)
            if fe < g.row_end:
                g.sweep(0, g.row_end - 2, g.row_end)
                g.sweep_next(g.row_end - 1, g.row_end)
            g.iterate_rows(fb, fe)
        g.finish_iterations()
        g.synchronize()

    def energy(self):
        return self.g.energy()


def spawn_ranks(n):
    """Run this script as n child processes (ranks 0..n-1, one per GPU) with the environment
    torch.distributed.run would give them; returns the first non-zero exit status, else 0."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    try:
        pending = list(children)
        while pending and status == 0:
            time.sleep(0.2)
            for child in list(pending):
                rc = child.poll()
                if rc is not None:
                    pending.remove(child)
                    status = status or rc
    finally:
        for child in children:  # a rank that failed must not leave its peers waiting for it
            if child.poll() is None:
                child.terminate()
        for child in children:
            try:
                child.wait(timeout=30)
            except subprocess.TimeoutExpired:
                child.kill()
    return status


def kernel_account(graph, steps, dtype, D, H, W, workload, world):
    """After profile_end: this rank's launches of the timed region by kind and the `roofline`
    object of its dominant kernel (algorithmic bytes of that kernel's launches / the HIP-event
    time of those launches)."""
    rb, re = graph.row_begin, graph.row_end
    elem = 4 if dtype == "f32" else 8
    my_pl = valid_pairs(re - rb, W, D)
    tname = "float" if dtype == "f32" else "double"
    lpp = 16 if D <= 64 else 32 if D <= 128 else 64
    n = [0, 0, 0]
    ms = [0.0, 0.0, 0.0]
    for kind in range(3):
        n[kind], ms[kind] = graph.profile_kind(kind)
    f2 = graph.fused2_info()
    n2, ms2 = n[2], ms[2]
    if n2:
        # Two iterations per launch over the interior: per (pixel, label) of that region and per
        # launch, 4 rows of the second colour read + 4 written (4*sizeof(T), half the pixels are
        # of that colour) + the cost byte of both iterations.
        bytes_per_pl = 4 * elem + 2
        kernel = "ds::fused2_iteration_kernel<%s, %d, %d, %d>" % (tname, lpp, 12 if dtype == "f32" else 8,
                                                                  0 if D == 4 * lpp else 1 if D % 4 == 0 else 2)
        kernel_s, launches, per_launch = ms2 / 1e3, n2, 2
        # valid (pixel, label) pairs of the interior region: column j has min(j, D - 1) + 1 labels.
        # A step covers the region once: in one launch, or (deep-halo bands) in up to three
        # launches over disjoint rows, so the units of a launch are the region's over launches per step
        per_step = n2 / max(steps // 2, 1)
        units_per_launch = (sum(min(j, D - 1) + 1 for j in range(f2["cols"][0], f2["cols"][1]))
                            * (f2["rows"][1] - f2["rows"][0])) / per_step
        note = ("one launch = TWO iterations over the interior (columns %d-%d, rows %d-%d); it moves "
                "(4*sizeof(T)+2)/2 B per valid pixel*label*iteration and is bound by the vector ALU, not by "
                "HBM: frac is what the metric asks for (algorithmic bytes / time / HBM peak), "
                "frac_at_one_iteration_bytes prices the same time with the one-iteration kernel's "
                "4*sizeof(T)+1 B per iteration, frac_at_two_pass_bytes with the half-sweeps' 8*sizeof(T)+1 B. "
                "The frame around the interior runs through the one-iteration kernel beside it "
                "(other_launches)" % (f2["cols"] + f2["rows"]))
    elif n[1]:
        # Whole iterations in one kernel: only the second colour's phi moves (4 rows read, 4 written
        # per pixel of that colour = 4*sizeof(T) per (pixel, label)) plus every pixel's cost byte.
        bytes_per_pl = 4 * elem + 1
        kernel = "ds::fused_iteration_kernel<%s, %d, %d, false>" % (tname, lpp, 16 if dtype == "f32" else 8)
        kernel_s, launches, per_launch = ms[1] / 1e3, n[1], 1
        # one launch = one iteration over the rows the kernel updates: all of them on one GPU, all
        # but the rows next to a neighbouring band otherwise (those go through half-sweeps)
        fb, fe = graph.fused_rows()
        whole = rb == 0 and re == H
        units_per_launch = my_pl if whole else my_pl * (fe - fb) / float(re - rb)
        note = ("one launch = one whole iteration that moves only the second colour's phi (4*sizeof(T)+1 B per "
                "valid pixel*label); priced with the two half-sweeps' 8*sizeof(T)+1 B the same time reads "
                "frac_at_two_pass_bytes")
    else:
        bytes_per_pl = 8 * elem + 1       # each phi row read once + written once, u8 cost read once
        kernel = "ds::sweep_kernel<%s, %d, false>" % (tname, lpp)
        kernel_s, launches, per_launch = ms[0] / 1e3, n[0], 0.5
        units_per_launch = my_pl * steps * 1.0 / max(launches, 1)  # a colour (or part of it) per launch
        note = "one launch = one colour's half-sweep"
    achieved = bytes_per_pl * units_per_launch * launches / kernel_s / 1e9 if kernel_s > 0 else 0.0
    avg_launch_s = kernel_s / max(launches, 1)
    pmc_bytes = pmc_traffic(workload, kernel) if world == 1 else None
    scale = (2 if n2 else 1) * units_per_launch * launches / kernel_s / 1e9 / HBM_PEAK_GBS if kernel_s > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_bytes / avg_launch_s / 1e9 if pmc_bytes and avg_launch_s > 0 else None,
                "traffic_bytes_per_launch": pmc_bytes,
                "kernel": kernel,
                "algorithmic_bytes_per_pixel_label": bytes_per_pl,
                "iterations_per_launch": per_launch,
                "two_pass_bytes_per_pixel_label": 8 * elem + 1,
                "frac_at_two_pass_bytes": (8 * elem + 1) * scale,
                **({"frac_at_one_iteration_bytes": (4 * elem + 1) * scale} if n2 else {}),
                "note": note,
                "launches": launches,
                "avg_launch_ms": avg_launch_s * 1e3,
                "band_halo": os.environ.get("DS_BAND_HALO", "one-row") if world > 1 else None,
                "other_launches": {"half_sweeps": n[0], "half_sweeps_ms": ms[0],
                                   "one_iteration_launches": n[1] if n2 else 0,
                                   "one_iteration_ms": ms[1] if n2 else 0.0} if (n[1] or n2) else None,
                "placement": dict(graph.placement_info(), mode=os.environ.get("DS_PLACEMENT", "default"))}
    return n, ms, roofline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="4k_256", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true",
                    help="N = 1: leave out the sustained run and the f64 run that follow the timed steps")
    ap.add_argument("--sustained-steps", type=int, default=400)
    ap.add_argument("--band-of", type=int, default=1,
                    help="with --gpus 1: time only band 0 of this many row bands, without halo exchange "
                         "(what one GPU of a larger job sweeps; a timing, not a result)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly (not by torch.distributed.run): start one rank per GPU ourselves.
        # This process has made no GPU call yet and makes none: it only waits for its children.
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # Rehearsal on a one-GPU box (DS_BENCH_REHEARSAL=1): all ranks share device 0 and their halo
    # rows travel through host memory, because RCCL refuses two ranks on one device.  Never
    # used for reported numbers.
    rehearsal = os.environ.get("DS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # gloo is the control plane only (the communicator's id, barriers, the max over ranks);
        # the halo rows travel over RCCL inside the library
        # (gloo and RCCL report on stdout when they connect: keep that away from the one JSON line)
        with stdout_to_stderr():
            dist.init_process_group("gloo")
            dist.barrier()

    def all_agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    wl = WORKLOADS[args.workload]
    H, W, D, dtype, lam = wl["H"], wl["W"], wl["D"], wl["dtype"], 1.2
    left, right, _ = synth.synthetic_pair(H, W, D)
    rb, re = bands.partition_rows(H, world)[rank]
    if args.band_of > 1:
        if world != 1:
            raise SystemExit("--band-of is a one-GPU measurement")
        rb, re = bands.partition_rows(H, args.band_of)[0]
    graph = diffstereo.DiffusionGraph(left, right, D, lam, dtype=dtype, row_begin=rb, row_end=re,
                                      device=local_rank)
    transport, comm = "none", None
    if world == 1:
        # the whole image through ds_iterate; --band-of: the band's kernels without a neighbour
        runner = SingleGpu(graph)
    else:
        if not rehearsal:
            # Preflight, in steps every rank can fail at without leaving its peers inside a
            # collective: can librccl be loaded; the communicator; one halo exchange.  If any rank
            # fails, all ranks move the halo rows through host memory over gloo instead (slower,
            # and said so in the output line) rather than losing the run.
            probe = (C.c_uint8 * diffstereo.DS_COMM_ID_BYTES)()
            ok = diffstereo.lib().ds_comm_unique_id(probe) == 0
            if all_agree(ok):
                try:
                    with stdout_to_stderr():
                        comm = bands.RcclComm(rank, world, local_rank)
                        runner = bands.Bands(graph, comm, rank, world)
                        runner.exchange_all()
                except Exception as exc:  # noqa: BLE001
                    ok = False
                    print("rank %d: RCCL halo exchange failed: %r" % (rank, exc), file=sys.stderr, flush=True)
                ok = all_agree(ok)
            if ok:
                transport = comm.name
            else:
                comm = None
        if comm is None:
            host = bands.GlooTransport()
            runner = bands.Bands(graph, host, rank, world)
            runner.exchange_all()
            transport = ("REHEARSAL: all ranks on one GPU, " if rehearsal else "RCCL failed in the preflight: ") + host.name

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(run, g, steps, warmup):
        """W warm-up steps, then `steps` steps between two fences; (seconds, launches timed)."""
        # First use, part of the set-up like ds_create: the two-iteration path allocates its third
        # set of planes, chooses its row segments and loads its kernels at its first step.  Two
        # iterations bring that about whatever --warmup says; the W warm-up steps follow.
        run.iterate(2)
        fence()
        run.iterate(warmup)
        fence()
        # launches per step: a band has boundary half-sweeps + the interior kernel; a two-iteration
        # step has the interior kernel + the launches for the frame around it
        g.profile_begin(steps * 8 + 16)
        t0 = time.perf_counter()
        run.iterate(steps)
        fence()
        dt = time.perf_counter() - t0
        return dt, g.profile_end()

    my_elapsed, prof = timed(runner, graph, args.steps, args.warmup)
    elapsed = my_elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    energy = runner.energy()
    # every rank's own account: its launches by kind (HIP events on the stream each launch is queued
    # on: the interior kernel on the handle's stream, the frame's on the frame stream, the boundary
    # rows' half-sweeps on the boundary stream -- the small ones include the time they wait for a
    # compute unit the interior kernel holds) and what its wall time holds beyond the interior kernel
    n_kind, ms_kind, my_roofline = kernel_account(graph, args.steps, dtype, D, H, W, args.workload, world)
    per = 1.0 / max(args.steps, 1)
    interior_ms = (ms_kind[2] if n_kind[2] else ms_kind[1]) * per
    account = {"rank": rank, "rows": [rb, re], "wall_ms_per_step": my_elapsed * per * 1e3,
               "interior_kernel_ms_per_step": interior_ms,
               "interior_kernel": my_roofline["kernel"],
               "two_iteration_launches": n_kind[2],
               "frame_launches_ms_per_step": (ms_kind[1] * per) if n_kind[2] else 0.0,
               "boundary_half_sweeps_ms_per_step": ms_kind[0] * per,
               "exposed_beyond_interior_kernel_ms_per_step": my_elapsed * per * 1e3 - interior_ms,
               "launches": prof["launches"], "roofline": my_roofline}
    per_rank = [account]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, account)

    line = None
    if rank == 0:
        pl = valid_pairs(H, W, D) if args.band_of == 1 else valid_pairs(re - rb, W, D)
        slowest = max(per_rank, key=lambda a: a["wall_ms_per_step"])
        roofline = dict(slowest["roofline"], rank=slowest["rank"])
        line = {
            "metric": "G pixel*label*iters/s (node), 4Kx256 disparities",
            "value": pl * args.steps / elapsed / 1e9,
            "unit": "G pixel*label*iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "synthetic %dx%d stereo pair, %d disparities, lambda=1.2, %s phi, "
                                   "u8 cost volume in HBM, red-black diffusion iterations" % (W, H, D, dtype),
                       "valid_pixel_labels": pl, "row_bands": world,
                       "halo_exchange": transport,
                       **({"partial": "band 0 of %d (rows %d-%d) on one GPU, no halo exchange: timing only"
                           % (args.band_of, rb, re)} if args.band_of > 1 else {})},
            "roofline": roofline,
            "energy": energy,
        }
        if world > 1:
            line["per_rank"] = [{k: v for k, v in a.items() if k != "roofline"} for a in per_rank]

    companions = world == 1 and args.band_of == 1 and not args.no_companions
    if companions and args.sustained_steps > 0:
        # the same handle for a few seconds: clock droop of a kernel bound by the vector ALU shows here
        t0 = time.perf_counter()
        runner.iterate(args.sustained_steps)
        fence()
        dt = time.perf_counter() - t0
        line["sustained"] = {"steps": args.sustained_steps, "ms_per_step": dt / args.sustained_steps * 1e3,
                             "value": valid_pairs(H, W, D) * args.sustained_steps / dt / 1e9, "seconds": dt}
    graph.close()
    if companions and dtype == "f32" and (args.workload + "_f64") in WORKLOADS:
        # the same workload in the reference's own precision
        g64 = diffstereo.DiffusionGraph(left, right, D, lam, dtype="f64")
        dt, _ = timed(SingleGpu(g64), g64, args.steps, args.warmup)
        _, _, roof64 = kernel_account(g64, args.steps, "f64", D, H, W, args.workload + "_f64", 1)
        line["f64"] = {"ms_per_step": dt / args.steps * 1e3, "value": valid_pairs(H, W, D) * args.steps / dt / 1e9,
                       "steps": args.steps, "energy": g64.energy(), "roofline": roof64}
        g64.close()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(left, right, D, lam)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
