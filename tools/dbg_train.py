import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_train import build
from conftest import rel_err
msg, spec, data, eng, ref, rsolver, rng = build(lr=0.0)
for k, v in data.items(): eng.host_array(k)[...] = v
out = eng.step(seed=7)
ref.blobs.update(data); ref.dropout_seed = 7; ref.forward(); grads = ref.backward()
names = [l.tops[0] for l in spec.layers if l.type in ("Convolution","Pooling","LRN","Dropout") and l.tops[0] in eng.grad_blobs]
for name in reversed(names):
    if name.endswith('/output'): continue
    if name in ref.diffs:
        g, r = eng.read_grad(name), ref.diffs[name]
        e = rel_err(g, r)
        l2 = float(np.linalg.norm((g - r).astype(np.float64)) / max(np.linalg.norm(r.astype(np.float64)), 1e-30))
        nbad = int((np.abs(g - r) > 1e-3 * np.abs(r).max()).sum())
        print("%-32s max %.3e  l2 %.3e  bad %d / %d %s" % (name, e, l2, nbad, g.size, "<<<" if e > 1e-3 else ""))
got = eng.download_grads()
for name, gs in grads.items():
    for i,(g, r) in enumerate(zip(got[name], gs)):
        e = rel_err(g, r)
        l2 = float(np.linalg.norm((g - r).astype(np.float64)) / max(np.linalg.norm(r.astype(np.float64)), 1e-30))
        if e > 1e-3: print("PARAM", name, i, "max %.2e l2 %.2e" % (e, l2))
