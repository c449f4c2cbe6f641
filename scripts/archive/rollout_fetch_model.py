"""Why FETCH_SIZE says 528 GB for rollout_bordered_kernel while the algorithm streams 895 GB (config 5 share:
8 series x 10,000 paths x 256 steps).  Every row read of the per-sample factor is ceil(a/4) lanes x 16 B, fetched in
128-byte lines (profiles/r02/a_rollout_rdreq_sizes.json: 8.25e9 read requests, ALL of the 128-B kind); FETCH_SIZE
tallies each request at 64 B.  Model: row a (a floats) is read at steps idx = a+1 .. H-1."""
H, paths = 256, 8 * 10000
alg = true = counted = 0
for a in range(1, H):
    w = H - 1 - a
    b = 16 * ((a + 3) // 4)                 # bytes the lanes ask for
    lines = (b + 127) // 128                # 128-B lines fetched (rows are 1 KiB apart: no line is shared)
    alg += 4 * a * w
    true += 128 * lines * w
    counted += 64 * lines * w
print(f"algorithmic {alg * paths / 1e9:.1f} GB   fetched (128-B lines) {true * paths / 1e9:.1f} GB   "
      f"FETCH_SIZE would read {counted * paths / 1e9:.1f} GB   fetched/algorithmic {true / alg:.3f}")
