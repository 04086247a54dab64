import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k], b[k]
    print(f"{k:8s} max|a-b| {np.abs(x - y).max():.3e}  max|a| {np.abs(x).max():.3e}  rel {np.abs(x - y).max() / max(np.abs(x).max(), 1e-30):.3e}")
for k in ("fb_y", "fb_g0", "ag_y", "ag_g0", "ag_w1", "fb_w1"):
    for nm, d in (("A", a), ("B", b)):
        print(f"{nm} run-to-run {k}: {np.abs(d[k + '0'] - d[k + '1']).max():.3e}")
print("A fb vs ag g0:", np.abs(a["fb_g00"] - a["ag_g00"]).max() / np.abs(a["fb_g00"]).max(), " B:", np.abs(b["fb_g00"] - b["ag_g00"]).max() / np.abs(b["fb_g00"]).max())
print("A fb vs ag y:", np.abs(a["fb_y0"] - a["ag_y0"]).max(), " B:", np.abs(b["fb_y0"] - b["ag_y0"]).max())
