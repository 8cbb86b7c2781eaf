import numpy as np, os, sys
root = sys.argv[1]
os.makedirs(root, exist_ok=True)
rs = np.random.RandomState(0)
names = []
for i in range(6):
    depth = (rs.rand(240, 320) * 3000 + 1000).astype(np.float32)
    kp = np.concatenate([rs.rand(20, 1) * 300 + 10, rs.rand(20, 1) * 220 + 10, rs.rand(20, 1)], axis=1)
    np.savez(os.path.join(root, 'f%d.npz' % i), depth=depth, keypoints=kp)
    names.append('f%d.npz' % i)
open(os.path.join(root, 'train.txt'), 'w').write('\n'.join(names) + '\n')
