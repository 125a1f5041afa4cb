"""How fast this GPU serves independent random reads of 32-byte and 64-byte records, the access
pattern of the traversal kernels (one tree record or one primitive record per lane and trip), for
tables that fit the L2s, the memory-side cache, or neither; and a plain device copy for comparison.
torch's index_select is the gather kernel (every row read by adjacent lanes, nothing else computed),
so the numbers are an upper bound the traversal loop, which has to compute between its reads and
chains them (a read's address comes out of the one before), cannot reach.

usage: python tools/gather_ceiling.py"""
import time

import torch


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps


def main():
    dev = torch.device("cuda:0")
    n = 1 << 30
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    s = timed(lambda: b.copy_(a))
    print("device copy of 4 GiB: %.0f GB/s read + written" % (2 * 4 * n / s / 1e9))
    del a, b
    m = 1 << 24
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for row_bytes in (32, 64):
        for table_mb in (8, 64, 192, 750, 4096):
            rows = table_mb * (1 << 20) // row_bytes
            table = torch.zeros((rows, row_bytes // 4), dtype=torch.float32, device=dev)
            idx = torch.randint(0, rows, (m,), device=dev, generator=g)
            out = torch.empty((m, row_bytes // 4), dtype=torch.float32, device=dev)
            s = timed(lambda: torch.index_select(table, 0, idx, out=out), reps=20)
            print("random %d-byte records from a %4d MB table: %6.1f G records/s, %6.0f GB/s gathered" % (row_bytes, table_mb, m / s / 1e9, m * row_bytes / s / 1e9), flush=True)
            del table, idx, out


if __name__ == "__main__":
    main()
