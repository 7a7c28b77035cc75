"""Round-6 probe: two dumps of dump_onepass_state.py (good build, bad build) compared: slab maxima / sums per head, e rows per block and token."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])   # a = good, b = bad
np.set_printoptions(linewidth=250, precision=4, suppress=True)
mg, mb = a["ws_ml"], b["ws_ml"]
print("head 3 (bh=3), slabs 0..9: m good", mg[:10, 3, 0], "\n                           m bad ", mb[:10, 3, 0])
print("                           l good", mg[:10, 3, 1], "\n                           l bad ", mb[:10, 3, 1])
print("head 1 (bh=1) m good", mg[:6, 1, 0], "bad", mb[:6, 1, 0])
x, y = a["sc"].view(np.float16).astype(np.float32).reshape(256, -1), b["sc"].view(np.float16).astype(np.float32).reshape(256, -1)
ld = x.shape[1]
xg, yg = x.reshape(64, 4 * ld), y.reshape(64, 4 * ld)
e_x, e_y = xg[:, :124 * 256].reshape(64, 124, 4, 64), yg[:, :124 * 256].reshape(64, 124, 4, 64)
for h in (1, 3):
    r = np.median(e_y[0, :16, h] / np.maximum(e_x[0, :16, h], 1e-3), axis=-1)
    print("group 0 head", h, "blocks 0..15: median e bad/good", r, " => m_used(bad) - m_used(good) =", -np.log(r))
    print("     max e per block good", e_x[0, :16, h].max(-1), "\n     max e per block bad ", e_y[0, :16, h].max(-1))
for blk in (0, 1, 2, 3, 6):
    r = e_y[0, blk, 3] / np.maximum(e_x[0, blk, 3], 1e-3)
    print("block", blk, "head 3 ratio per token:\n", r)
# how many (group, block) pairs are affected at all, by position of the block inside its workgroup (block % 4)
d = (np.abs(e_x - e_y).max(-1) > 0)      # [64 groups, 124 blocks, 4 heads]
print("affected (group, block) count per head:", d.sum((0, 1)), " by block % 4 (head 3):", [int(d[:, k::4, 3].sum()) for k in range(4)], "of", d[:, 0::4, 3].size)
print("affected workgroups per group (head 3):", d[:, :, 3].reshape(64, 31, 4).any(-1).sum(-1))
