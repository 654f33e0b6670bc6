import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
worst = (0.0, '')
for k in a.files:
    e = np.abs(a[k] - b[k]); m = np.abs(a[k]).max() + 1e-30
    bad = e > 1e-4 * m
    line = f"{k:70s} {str(a[k].shape):22s} rel max {e.max() / m:.2e} frac>1e-4 {bad.mean():.5f}"
    if bad.any() and a[k].ndim == 4:
        # which output / input channels carry the differences (conv weights are [a, b, 4, 4])
        ia = np.where(bad.any(axis=(1, 2, 3)))[0]; ib = np.where(bad.any(axis=(0, 2, 3)))[0]
        line += f" | axis0 bad {len(ia)}/{a[k].shape[0]} [{ia.min()}..{ia.max()}] axis1 bad {len(ib)}/{a[k].shape[1]} [{ib.min()}..{ib.max()}]"
    elif bad.any() and a[k].ndim == 1:
        i = np.where(bad)[0]; line += f" | bad idx {i[:12].tolist()}"
    print(line)
    worst = max(worst, (float(e.max() / m), k))
print('WORST %.3e %s' % worst)
