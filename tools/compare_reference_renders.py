"""the only radiometric evidence the reference itself carries are the renders under /root/reference/img (parameters
unrecorded -- LOOSE).  tools/render_for_compare.py rendered the matching scene of main.rs at each stored image's resolution on the GPU box
(PNGs under gpurun_out/); this script, on the CPU where the reference is readable, reports per pair
  * geometry-mask IoU: both images segmented by the SAME rule into black (every channel < 10: the `no_light` background), saturated
    (every channel > 245: light sources seen directly) and the rest; IoU per class and their mean over the classes either image has;
  * luminance correlation: Pearson r of Rec.709 luminance after a 16x16 box filter (robust to Monte Carlo noise and to texture detail);
  * mean luminance of each.
into profiles/r03_reference_renders.json.  It stays "parity unpinned": a differing scene version, camera or asset shows up as a low score,
and the note of each pair says what is known to differ.  usage: compare_reference_renders.py [gpurun_out dir] [reference img dir]"""
import json, os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
ref_dir = sys.argv[2] if len(sys.argv) > 2 else "/root/reference/img"
meta = json.load(open(os.path.join(src, "r03_cmp_meta.json")))


def classes(a):
    return np.where((a < 10).all(axis=2), 0, np.where((a > 245).all(axis=2), 1, 2))


def lum(a):
    return a[..., 0] * 0.2126 + a[..., 1] * 0.7152 + a[..., 2] * 0.0722


def box(a, k=16):
    h, w = a.shape[0] // k * k, a.shape[1] // k * k
    return a[:h, :w].reshape(h // k, k, w // k, k).mean(axis=(1, 3))


out = []
for m in meta:
    ours = np.asarray(Image.open(os.path.join(src, f"r03_cmp_{m['tag']}.png")).convert("RGB"), np.float64)
    ref_im = Image.open(os.path.join(ref_dir, os.path.basename(m["reference_image"]))).convert("RGB")
    assert ref_im.size == (m["width"], m["height"]), (ref_im.size, m["width"], m["height"])
    ref = np.asarray(ref_im, np.float64)
    co, cr = classes(ours), classes(ref)
    iou = {}
    for k, name in enumerate(("black_background", "saturated_light", "lit_surfaces")):
        u = ((co == k) | (cr == k)).sum()
        if u:
            iou[name] = float(((co == k) & (cr == k)).sum() / u)
    lo, lr = box(lum(ours)), box(lum(ref))
    r = float(np.corrcoef(lo.ravel(), lr.ravel())[0, 1])
    out.append({**m, "mask_iou": iou, "mask_iou_mean": float(np.mean(list(iou.values()))), "luminance_correlation_16x16": r,
                "mean_luminance_ours": float(lum(ours).mean()), "mean_luminance_reference": float(lum(ref).mean()),
                "class_share_ours": [float((co == k).mean()) for k in range(3)], "class_share_reference": [float((cr == k).mean()) for k in range(3)]})
    print(f"{m['tag']:16s} IoU {out[-1]['mask_iou_mean']:.3f} {iou}  luminance r {r:.3f}  mean lum {out[-1]['mean_luminance_ours']:.1f} / {out[-1]['mean_luminance_reference']:.1f}")
res = {"caveat": "parameters of the stored renders are unrecorded (camera, spp, scene version, assets) -- loose evidence, parity stays unpinned",
       "method": __doc__.split("reports per pair")[1].split("into profiles")[0].strip(), "pairs": out}
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_reference_renders.json"), "w"), indent=1)
