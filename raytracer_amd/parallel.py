"""Multi-GPU frame rendering: one process per GPU, sample split, one reduce.

Samples and pixels of a frame are independent, so the path shards without any exchange during
rendering: every rank holds the whole scene, renders a disjoint range of sample indices of one
global stratified grid into its own radiance-SUM image, and the images are added onto rank 0 with a
single `torch.distributed.reduce` (RCCL over xGMI on a GPU node, gloo in the CPU tests).  Because
the sample streams are counter-based (keyed by seed, pixel and sample index), the union of the
ranks' samples is exactly the sample set a single GPU would draw for the same grid.

A tile split is the other way to divide ONE frame (strong scaling): rank r renders ALL samples of the
8x8-pixel tiles (tx, ty) with (tx + ty) % world == r (`tile_share`).  The shares interleave across
the image, so they balance without stealing; every rank still leaves the other ranks' pixels at zero
in a full-size image, so the same SUM reduce combines them (adding zeros is exact: the frame equals the
one-GPU frame pixel for pixel when both render a pixel's samples in one pass).  It is the faster split
where per-pixel costs matter: a GPU that renders 1/8 of the pixels at 1024 samples each traces its
camera rays at the 1024-sample rate, one that renders all pixels at 128 samples each pays the per-pixel
cache misses of the whole image (measured on the bench frame: bench.py, strong_proxy).  It needs an image
width that is a multiple of 8 * world; `rank_params` falls back to the sample split otherwise.
Disjoint scanline ranges (`row_range`) remain available; they are not load balanced.
"""
from .host import render_params


def sample_range(rank, world, n_samples):
    """Contiguous, disjoint, exhaustive split of [0, n_samples) (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(n_samples, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def row_range(rank, world, height):
    return sample_range(rank, world, height)


def tile_share_possible(width, height, world):
    """Whether mrt_render accepts tile shares of `world` ranks for this image size."""
    return world > 1 and width % (8 * world) == 0 and height % 8 == 0


def rank_params(width, height, spp_x, spp_y, rank, world, max_depth=5, seed=1, weak=False, split="sample"):
    """Render parameters of one rank.

    weak=False, split="sample": the spp_x * spp_y samples of the frame are divided among the ranks.
    weak=False, split="tile"  : the frame's 8x8 tiles are dealt to the ranks, all samples each (falls back
                                to the sample split where the image size does not allow it).
    weak=True                 : every rank renders spp_x * spp_y samples of a spp_x x (spp_y * world) grid.
    """
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    if weak:
        spp = spp_x * spp_y
        return render_params(width, height, spp_x, spp_y * world, max_depth=max_depth, seed=seed,
                             sample_begin=rank * spp, sample_end=(rank + 1) * spp)
    if split == "tile" and tile_share_possible(width, height, world):
        return render_params(width, height, spp_x, spp_y, max_depth=max_depth, seed=seed, tile_step=world, tile_phase=rank)
    b, e = sample_range(rank, world, spp_x * spp_y)
    return render_params(width, height, spp_x, spp_y, max_depth=max_depth, seed=seed, sample_begin=b, sample_end=e)


def render_frame(render, params, image, dist=None, dst=0):
    """Renders this rank's share into `image` (radiance sums, a torch tensor) and reduces onto `dst`.

    render(params, image) adds the rank's samples to `image`; on a GPU it is DeviceScene.render.
    Returns whatever `render` returned (the rank's own statistics).
    """
    image.zero_()
    stats = render(params, image)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return stats
