import sys
sys.path.insert(0, 'ab/w16')
from banks import rd128, rd64, wr64, wr128

print("== weight image 64 cols (W2): stride el -> fwd b128 extra, fwd as 2 x b64 extra (each), tr extra")
for LD in range(64, 100, 2):
    if LD % 4: continue
    f128 = rd128([2 * ((l & 15) * LD + 8 * (l >> 4)) for l in range(64)]) if LD % 8 == 0 else None
    f64a = rd64([2 * ((l & 15) * LD + 8 * (l >> 4)) for l in range(64)])
    f64b = rd64([2 * ((l & 15) * LD + 8 * (l >> 4) + 4) for l in range(64)])
    tr = rd64([2 * ((4 * (l >> 4) + ((l & 15) >> 2)) * LD + 8 * (l & 3)) for l in range(64)])
    print(LD, f128, f64a, f64b, tr)
print("== wave image: stride -> st b128 extra, dW32 tr (rows 4q+c) extra, (rows q+4c), tail16 tr")
for LD in range(64, 100, 4):
    st = wr128([2 * ((l & 15) * LD + 8 * (l >> 4)) for l in range(64)]) if LD % 8 == 0 else None
    st64 = wr64([2 * ((l & 15) * LD + 4 * (l >> 4)) for l in range(64)])
    res = []
    for rowf in (lambda q, c: 4 * q + c, lambda q, c: q + 4 * c):
        w = 0
        for rd in (0, 1):
            w = max(w, rd64([2 * (rowf((l & 15) >> 2, 2 * (l >> 5) + rd) * LD + 16 * ((l >> 4) & 1) + 4 * (l & 3)) for l in range(64)]))
        res.append(w)
    print(LD, st, st64, res)
