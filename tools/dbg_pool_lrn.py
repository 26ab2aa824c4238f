import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from fcn_object_detector_amd import lib as L
from gpu_util import dev_from, dev_to
from oracle import caffe_ref as R
n,h,w,c = 16,45,52,96
rng = np.random.default_rng(n*h+c)
x = (rng.standard_normal((n, h, w, c)) * 30 - 5).astype(np.float16)
oh, ow = R.pool_out(h, 3, 0, 2), R.pool_out(w, 3, 0, 2)
xd = dev_from(x); yd = dev_from(np.zeros((n, oh, ow, c), np.float16))
L.call("fcn_maxpool_lrn5_fwd_f16", xd.ptr, yd.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 1, 1e-4, 0.75, 1.0, None)
md = dev_from(np.zeros((n, h, w, c), np.float16)); zd = dev_from(np.zeros((n, oh, ow, c), np.float16))
L.call("fcn_lrn_fwd_f16", xd.ptr, md.ptr, n * h * w, c, c, c, 5, 1e-4, 0.75, 1.0, None)
L.call("fcn_maxpool_fwd_f16", md.ptr, zd.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 0, None)
a = dev_to(yd, (n, oh, ow, c), np.float16); b = dev_to(zd, (n, oh, ow, c), np.float16)
bad = np.argwhere(a != b)
print("mismatches", len(bad), "of", a.size, "oh,ow", oh, ow)
print(bad[:10]); 
if len(bad):
    i = tuple(bad[0]); print(a[i], b[i])
    print("by image", np.bincount(bad[:,0], minlength=n)); print("by oy", np.bincount(bad[:,1], minlength=oh)); print("by ox", np.bincount(bad[:,2], minlength=ow)); print("by c//8", np.bincount(bad[:,3]//8, minlength=12))
outs = []
for rep in range(4):
    yd2 = dev_from(np.zeros((n, oh, ow, c), np.float16))
    L.call("fcn_maxpool_lrn5_fwd_f16", xd.ptr, yd2.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 1, 1e-4, 0.75, 1.0, None)
    outs.append(dev_to(yd2, (n, oh, ow, c), np.float16))
print("run-to-run differences:", [int((outs[0] != o).sum()) for o in outs[1:]], "vs two-launch:", [int((o != b).sum()) for o in outs])
# which side is right?  float64 LRN of the mismatching positions
x32 = x.astype(np.float64)
def lrn_ref(img, yy, xx, ch):
    px = x32[img, yy, xx]
    lo, hi = max(ch - 2, 0), min(ch + 3, c)
    return px[ch] * (1.0 + 1e-4 / 5 * (px[lo:hi] ** 2).sum()) ** -0.75
for (img, oy_, ox_, ch) in bad[:6]:
    vals = [lrn_ref(img, min(2*oy_+dy, h-1), min(2*ox_+dx, w-1), ch) for dy in range(3) for dx in range(3)]
    print((img, oy_, ox_, ch), "fused", float(a[img, oy_, ox_, ch]), "two-launch", float(b[img, oy_, ox_, ch]), "float64 max", max(vals), "as f16", float(np.float16(max(vals))))
