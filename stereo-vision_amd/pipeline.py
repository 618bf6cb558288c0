"""The reference's `main` (diffusion_stereo/src/main.rs:36-98) for one or many GPUs.

Left / right ASCII PGM in, disparity PGM out, with main's constants as defaults
(max disparity 40, smoothing 1.2, epsilon 1/(10*W*H), batches of 200 iterations):
diffusion batches until the epsilon-consistent sub-graph is non-empty
(crossing_out_graph.rs:254-275), then the smallest surviving label per pixel (:318-327).

One process per GPU.  Started by torch.distributed.run the ranks split the image into
row bands and the library runs the band protocol itself (ds_band_iterate, ds_band_energy,
ds_band_crossing_out: halo rows over RCCL, crossing-out with mask halos and reduced flags);
torch.distributed only carries the communicator's id and gathers the labels.  --backend gloo
moves the halo rows through host memory instead (ranks sharing one GPU).  Started plainly it
is the single-GPU pipeline.

    python stereo-vision_amd/pipeline.py LEFT.pgm RIGHT.pgm OUT.pgm [--max-disparity N]
        [--smoothing X] [--epsilon E] [--batch B] [--max-batches M] [--dtype f64|f32]
        [--backend nccl|gloo]
"""
import argparse
import os
import sys

import numpy as np


def pgm_reader(path):
    """pgm_handler.rs:36-74: line 1 magic (not checked), line 2 `W H`, line 3 maxval, then
    W*H whitespace-separated integers.  Returns (matrix uint32 [H][W], maxval)."""
    with open(path) as f:
        lines = f.read().split("\n")
    if len(lines) < 3:
        raise ValueError("%s: missing header lines" % path)
    width, height = (int(x) for x in lines[1].split()[:2])
    maxval = int(lines[2].split()[0])
    values = [int(tok) for line in lines[3:] for tok in line.split()]
    if len(values) != width * height:
        raise ValueError("%s: pixel count does not match width * height" % path)
    return np.asarray(values, dtype=np.int32).astype(np.uint32).reshape(height, width), maxval


def pgm_writer(matrix, path, max_intensity):
    """pgm_handler.rs:122-133: `P2`, `W H`, maxval, one space-joined row per line."""
    matrix = np.asarray(matrix)
    with open(path, "w") as f:
        f.write("P2\n%d %d\n%d\n" % (matrix.shape[1], matrix.shape[0], max_intensity))
        for row in matrix:
            f.write(" ".join(str(int(v)) for v in row) + "\n")


def check_disparity_map(disparity_map):
    """utils.rs:121-130: consistent iff d[i][j+1] <= d[i][j] + 1 along every row."""
    m = np.asarray(disparity_map, dtype=np.int64)
    return bool((m[:, 1:] <= m[:, :-1] + 1).all())


def diffusion_while_not_consistent(runner, epsilon, batch_size, max_batches=1000, log=None):
    """crossing_out_graph.rs:254-275 over the runner's bands.  Returns (iterations done,
    crossing-out result of this band); raises if max_batches batches do not suffice (the
    reference would loop forever).  Every rank must call it (the energy and the crossing-out
    flags are collectives); `log` may be None on the ranks that should stay silent."""
    done = 0
    for _ in range(max_batches):
        runner.iterate(batch_size)
        done += batch_size
        energy = runner.energy()  # a collective: on every rank, whether it logs or not
        if log:
            log("%d iterations of diffusion ended, energy %.17g" % (done, energy))
        result = runner.crossing_out(epsilon)
        if result["not_empty"]:
            if log:
                log("Graph is consistent. Diffusion is done (%d crossing-out passes)" % result["passes"])
            return done, result
        if log:
            log("Need more diffusion iterations")
    raise RuntimeError("graph is not consistent after %d batches" % max_batches)


def run(left, right, max_disparity=40, smoothing=1.2, epsilon=None, batch=200, max_batches=1000,
        dtype="f64", log=None, labeling="simple", transport=None, device=None):
    """The pipeline on already loaded images.  Uses the default process group if one is
    initialised (one band per rank), else a single band.  Returns (disparity map [H][W] on
    rank 0, None elsewhere; iterations done).  labeling: "simple" = simple_best_labeling
    (crossing_out_graph.rs:318-327, what the reference's main uses), "best" = find_best_labeling
    (:349-364; sequential over the whole image, so one band only).  transport: "rccl" (the
    library's own) or "gloo" (through host memory); default: by the process group's backend.
    device: the GPU's index (default: LOCAL_RANK modulo the number of GPUs, 0 in a single process).
    torch is the control plane only (the process group); its own GPU runtime is never started here."""
    import torch
    import torch.distributed as dist
    from . import bands, diffstereo
    H, W = left.shape
    if left.max() > 255 or right.max() > 255:
        raise ValueError("intensities must fit 8 bits")
    epsilon = 1.0 / (10.0 * W * H) if epsilon is None else epsilon
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if labeling not in ("simple", "best"):
        raise ValueError("labeling must be 'simple' or 'best'")
    if labeling == "best" and multi:
        raise ValueError("find_best_labeling walks the whole image pixel by pixel: run it as one process")
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    rb, re = bands.partition_rows(H, world)[rank]
    if device is None:  # device_count() asks the driver without starting torch's runtime
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1) if multi else 0
    graph = diffstereo.DiffusionGraph(left.astype(np.uint8), right.astype(np.uint8), max_disparity, smoothing,
                                      dtype=dtype, row_begin=rb, row_end=re, device=device)
    if transport is None:
        transport = "rccl" if multi and dist.get_backend() == "nccl" else "gloo"
    link = (bands.RcclComm(rank, world, device) if multi and transport == "rccl"
            else bands.GlooTransport())
    runner = bands.Bands(graph, link, rank, world)
    try:
        iters, result = diffusion_while_not_consistent(runner, epsilon, batch, max_batches,
                                                       log if rank == 0 else None)
        labels = result["labels"]
        if labeling == "best":
            best = graph.find_best_labeling()   # continues on the masks the crossing-out left
            if not best["not_empty"]:
                raise RuntimeError("Graph is empty")
            labels = best["labels"]
        if not multi:
            return labels, iters
        parts = [None] * world if rank == 0 else None
        dist.gather_object(labels, parts, dst=0)
        return (np.concatenate(parts, axis=0) if rank == 0 else None), iters
    finally:
        graph.close()
        link.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("left")
    ap.add_argument("right")
    ap.add_argument("out")
    ap.add_argument("--max-disparity", type=int, default=40)
    ap.add_argument("--smoothing", type=float, default=1.2)
    ap.add_argument("--epsilon", type=float, default=None)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--max-batches", type=int, default=1000)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--labeling", choices=("simple", "best"), default="simple")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    # ROCr reads this when the runtime starts, i.e. at the first HIP call of the process: set it
    # before anything can make one.  (The host driver only supports dmabuf IPC; without the
    # setting RCCL's buffer sharing between ranks fails.  A caller that has initialised the GPU
    # before calling main() must have exported it itself.)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")  # control plane; --backend chooses how halo rows travel
    rank = dist.get_rank() if world > 1 else 0
    try:
        right, _ = pgm_reader(a.right)
        left, _ = pgm_reader(a.left)
        if left.shape != right.shape:
            raise ValueError("left and right images differ in size")
        log = None if a.quiet else (lambda s: print(s, flush=True))
        labels, _ = run(left, right, a.max_disparity, a.smoothing, a.epsilon, a.batch, a.max_batches,
                        a.dtype, log, a.labeling, "rccl" if a.backend == "nccl" else "gloo")
        if rank == 0:
            print("Disparity map is consistent: %s" % ("true" if check_disparity_map(labels) else "false"))
            pgm_writer(labels, a.out, a.max_disparity)
            print("Disparity map is saved to `%s`" % a.out)
    finally:
        if world > 1:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    load_package()
    from stereo_vision_amd import pipeline as _self
    sys.exit(_self.main())
