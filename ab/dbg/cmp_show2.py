import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for i in range(11):
    print(i, "loss A", float(a[f"img_loss{i}"]), "B", float(b[f"img_loss{i}"]), "plan A", float(a[f"plan_loss{i}"]), "B", float(b[f"plan_loss{i}"]),
          "y max|A-B|", np.abs(a[f"img_y{i}"] - b[f"img_y{i}"]).max(), "g0", np.abs(a[f"img_g0{i}"] - b[f"img_g0{i}"]).max())
