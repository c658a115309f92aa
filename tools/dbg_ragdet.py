import sys, os, faulthandler
faulthandler.enable()
sys.path[:0] = [os.getcwd(), os.getcwd() + "/oracle", os.getcwd() + "/tools"]
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
rs = np.random.RandomState(41)
g = pkg.Net("det")
for sizes in ([(96, 160)], [(96, 160), (96, 160)], [(96, 160), (32, 32)], [(96, 160), (96, 160), (32, 32), (64, 224), (160, 96), (128, 128), (32, 96), (224, 64)]):
    imgs = [rs.randn(h, w, 3).astype(np.float32) for h, w in sizes]
    print("sizes", sizes, flush=True)
    y = g.forward_ragged_images(imgs, keep_all=2)
    print("ok", y.shape, flush=True)
