"""Multi-GPU frame rendering: one process per GPU, sample split, one reduce.

Samples and pixels of a frame are independent, so the path shards without any exchange during
rendering: every rank holds the whole scene, renders a disjoint range of sample indices of one
global stratified grid into its own radiance-SUM image, and the images are added onto rank 0 with a
single `torch.distributed.reduce` (RCCL over xGMI on a GPU node, gloo in the CPU tests).  Because
the sample streams are counter-based (keyed by seed, pixel and sample index), the union of the
ranks' samples is exactly the sample set a single GPU would draw for the same grid.

A tile split (disjoint scanline ranges, no arithmetic reduction) is available through
`row_range`; it needs a gather instead of a reduce and is not load balanced, so the sample split
is the default.
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


def rank_params(width, height, spp_x, spp_y, rank, world, max_depth=5, seed=1, weak=False):
    """Render parameters of one rank.

    weak=False: the spp_x * spp_y samples of the frame are divided among the ranks (strong scaling).
    weak=True : every rank renders spp_x * spp_y samples of a spp_x x (spp_y * world) grid.
    """
    if weak:
        spp = spp_x * spp_y
        return render_params(width, height, spp_x, spp_y * world, max_depth=max_depth, seed=seed,
                             sample_begin=rank * spp, sample_end=(rank + 1) * spp)
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
