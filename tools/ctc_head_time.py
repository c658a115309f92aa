"""Development script: the duration of the server recognizer's CTC head launch inside the pipeline (partial mode: csrc/srv_kernels.hip, GemmArgs::ctc_part)
at the cfg5 shapes - 32 images x 32 lines - from the pipeline's timing report (one chain).  python tools/ctc_head_time.py"""
import sys, os, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tools"]
import __graft_entry__ as ge
pkg = ge.load_package()
import synth_weights; synth_weights.ensure_server(ROOT)
from synth_data import cfg2_sample
n, hw, k = 32, 960, 32
samples = [cfg2_sample(300 + i, hw, hw, k) for i in range(n)]
srv = os.path.join(ROOT, "models_server")
pipe = pkg.Pipe(device=0, enable_cls=True, limit_side_len=hw, rec_batch_num=16, rec_img_h=48, rec_img_w=320, precision="fp16", phases=1,
                det_dir=os.path.join(srv, "det"), rec_dir=os.path.join(srv, "rec"))
d_i, d_p = pkg.DevArray(np.stack([s[0] for s in samples])), pkg.DevArray(np.stack([s[1] for s in samples]))
for _ in range(2): pipe.run_device(d_i, hw, hw, n, d_p)
pipe.timing(True)
for _ in range(3): pipe.run_device(d_i, hw, hw, n, d_p)
rep = pipe.timing_report()
for name, v in rep.items():
    if "6625" in name: print(name, round(v["ms"] / v["count"], 4), "ms x", v["count"])
pipe.close()
