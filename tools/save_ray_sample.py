"""Compact sample of the diagnostic build's ray log for offline hierarchy studies: save_ray_sample.py <log dir> <out.npz> [waves per launch].
Keeps complete 64-ray waves only: per launch an array [waves, 64, 6] of (origin, direction) in world space (fp32 as logged)."""
import glob, os, sys
import numpy as np
d, out = sys.argv[1], sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 256
res = {}
for g in sorted(glob.glob(os.path.join(d, "rays_*.bin"))):
    rays = np.fromfile(g, dtype=np.float32).reshape(-1, 8)
    ids = rays[:, 0].copy().view(np.uint32).astype(np.int64)
    order = np.argsort(ids); ids = ids[order]; rays = rays[order]
    wave = ids >> 6
    u, start, cnt = np.unique(wave, return_index=True, return_counts=True)
    full = np.flatnonzero(cnt == 64)
    if len(full) == 0:
        continue
    pick = full[np.linspace(0, len(full) - 1, min(per, len(full))).astype(np.int64)]
    res[os.path.basename(g)[:-4]] = np.stack([rays[start[k]:start[k] + 64, 1:7] for k in pick]).astype(np.float32)
np.savez_compressed(out, **res)
print({k: v.shape for k, v in res.items()})
