"""Lower a Paddle inference graph (`models/*/inference.pdmodel`) to this build's fused
layer table ("plan").  One plan line = one device kernel family invocation.

The plan is the build's own description of the three networks on the hot path
(reference: the graphs that `DBDetector/Classifier/CRNNRecognizer::LoadModel` hand to
Paddle Inference — /root/reference/src/ocr_det.cpp:23-91, ocr_cls.cpp:108-175,
ocr_rec.cpp:137-211; op semantics in SURVEY.md §A).  The generated text is committed
under `cpp-paddle-ocr_amd/plans/` and embedded into the C-ABI library; the oracle reads
the same text.  An independent, *unfused* interpretation of the pdmodel
(oracle/graph_ref.py, torch CPU) cross-checks both.

Plan grammar (one op per line, space separated `key=value`):
  conv   i= o= cin= cout= kh= kw= sh= sw= ph= pw= w=<filter>          [ep=...]
  dw     i= o= c= kh= kw= sh= sw= ph= pw= w=<filter>                  [ep=...]
  deconv i= o= cin= cout= w=<filter>  (k2 s2 p0)                      [ep=...]
  linear i= o= cin= cout= w=<[in,out] matrix>                         [ep=...]
  sefc   i= o= c= cr= w1= b1= w2= b2= slope= offset=    (relu between, hard-sigmoid after)
  gap    i= o= c=
  pool   i= o= c= type=avg|max kh= kw= sh= sw=           (no padding, exclusive)
  ew     i= o= c= ep=...
  concat i=a,b,.. up=s,s,.. o= c=
  ln     i= o= c= eps= g= b=
  attn   i= o= heads= hd= scale=          (input [N,T,3*heads*hd] -> [N,T,heads*hd])
  softmax i= o= c=
  output i=
Epilogue stages (`|` separated, applied left to right, each one rounding step(s) of f32):
  bias:<vec>            y = y + b[c]
  smul:<scalar>         y = a * y
  sadd:<scalar>         y = y + b
  bn:<g>,<b>,<m>,<v>,<eps>   y = y*s[c] + t[c], s = g*rsqrt(v+eps), t = b - m*s  (see csrc/epilogue.h)
  act:relu | act:hswish | act:hsig,<slope>,<offset> | act:swish | act:sigmoid
  mulc:<tid>            y = y * S[n,c]          (S is [N,1,1,C])
  addt:<tid>            y = y + T[n,h,w,c]
  addup:<tid>,<s>       y = y + T[n,h/s,w/s,c]  (nearest, align_corners=False)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pdmodel import Program  # noqa: E402

VIEW_OPS = {"flatten_contiguous_range", "transpose2", "reshape2", "squeeze2", "dropout", "assign"}
IGNORE_OPS = {"shape", "slice", "fill_constant", "feed"}
ACTS = {"relu": "relu", "hard_swish": "hswish", "swish": "swish", "sigmoid": "sigmoid"}


class PlanOp:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.kw = kw
        self.ep = []
        self.out_var = None  # current graph var this op's (fused) output corresponds to

    def line(self):
        parts = [self.kind]
        for k, v in self.kw.items():
            if isinstance(v, (list, tuple)):
                v = ",".join(str(x) for x in v)
            elif isinstance(v, float):
                v = repr(float(v))
            parts.append("%s=%s" % (k, v))
        if self.ep:
            parts.append("ep=" + "|".join(self.ep))
        return " ".join(parts)


def lower(prog):
    ops = prog.ops
    V = prog.vars

    def pers(n):
        return n in V and V[n]["persistable"]

    # consumer counts over "real" ops
    uses = {}
    for op in ops:
        if op.type in IGNORE_OPS and op.type != "feed":
            continue
        for k, args in op.inputs.items():
            for a in args:
                uses[a] = uses.get(a, 0) + 1

    plan = []
    tid = {}        # graph var -> tensor id
    chans = {}      # tensor id -> logical channel count
    producer = {}   # graph var -> PlanOp whose fused output it currently is
    upsampled = {}  # graph var -> (src var, scale)
    next_id = [0]

    def new_tid(c):
        t = next_id[0]
        next_id[0] += 1
        chans[t] = c
        return t

    def emit(kind, out_var, nch, **kw):
        t = new_tid(nch)
        p = PlanOp(kind, **kw)
        p.kw["o"] = t
        # keep key order: i first, o second
        kw2 = {}
        if "i" in p.kw:
            kw2["i"] = p.kw.pop("i")
        kw2["o"] = p.kw.pop("o")
        kw2.update(p.kw)
        p.kw = kw2
        p.out_var = out_var
        plan.append(p)
        tid[out_var] = t
        producer[out_var] = p
        return p

    def fusable(var):
        """var is the live output of a plan op, and nobody else reads it."""
        return var in producer and producer[var].out_var == var and uses.get(var, 0) == 1 \
            and producer[var].kind in ("conv", "dw", "deconv", "linear", "ew")

    def stage_on(var, out_var, stage):
        """Append `stage` to the op producing `var` (fusing) or open a new ew op."""
        if fusable(var):
            p = producer[var]
            p.ep.append(stage)
            p.out_var = out_var
            tid[out_var] = tid[var]
            producer[out_var] = p
        else:
            c = chans[tid[var]]
            p = emit("ew", out_var, c, i=tid[var], c=c)
            p.ep.append(stage)

    i = 0
    n = len(ops)
    while i < n:
        op = ops[i]
        t = op.type
        if t == "feed":
            x = op.out("Out")
            tid[x] = new_tid(3)
            i += 1
            continue
        if t in IGNORE_OPS:
            i += 1
            continue
        if t == "fetch":
            p = PlanOp("output", i=tid[op.inp("X")])
            plan.append(p)
            i += 1
            continue
        if t in ("conv2d", "depthwise_conv2d"):
            x = op.inp("Input")
            w = op.inp("Filter")
            wd = V[w]["dims"]
            st = op.attrs["strides"]
            pd = op.attrs["paddings"]
            assert op.attrs.get("dilations", [1, 1]) == [1, 1]
            assert len(pd) == 2
            g = op.attrs["groups"]
            cin = chans[tid[x]]
            # SE pattern: gap -> conv+bias+relu -> conv+bias+hsig
            if t == "conv2d" and x in producer and producer[x].kind == "gap" and wd[2] == 1:
                o1, o2, o3, o4, o5 = ops[i + 1], ops[i + 2], ops[i + 3], ops[i + 4], ops[i + 5]
                assert [o.type for o in (o1, o2, o3, o4, o5)] == \
                    ["elementwise_add", "relu", "conv2d", "elementwise_add", "hard_sigmoid"], \
                    [o.type for o in (o1, o2, o3, o4, o5)]
                w2 = o3.inp("Filter")
                emit("sefc", o5.out("Out"), cin, i=tid[x], c=cin, cr=wd[0], w1=w, b1=o1.inp("Y"),
                     w2=w2, b2=o4.inp("Y"), slope=op_attr_f(o5, "slope"), offset=op_attr_f(o5, "offset"))
                i += 6
                continue
            if t == "depthwise_conv2d" or (g == cin and g > 1 and wd[1] == 1):
                assert wd[0] == cin and wd[1] == 1 and g == cin
                emit("dw", op.out("Output"), cin, i=tid[x], c=cin, kh=wd[2], kw=wd[3],
                     sh=st[0], sw=st[1], ph=pd[0], pw=pd[1], w=w)
            else:
                assert g == 1 and wd[1] == cin, (wd, cin)
                emit("conv", op.out("Output"), wd[0], i=tid[x], cin=cin, cout=wd[0], kh=wd[2], kw=wd[3],
                     sh=st[0], sw=st[1], ph=pd[0], pw=pd[1], w=w)
            i += 1
            continue
        if t == "conv2d_transpose":
            x = op.inp("Input")
            w = op.inp("Filter")
            wd = V[w]["dims"]  # [cin, cout, 2, 2]
            assert wd[2:] == [2, 2] and op.attrs["strides"] == [2, 2] and op.attrs["paddings"] == [0, 0]
            assert op.attrs["groups"] == 1
            emit("deconv", op.out("Output"), wd[1], i=tid[x], cin=wd[0], cout=wd[1], w=w)
            i += 1
            continue
        if t == "matmul_v2" and pers(op.inp("Y")):
            x = op.inp("X")
            w = op.inp("Y")
            wd = V[w]["dims"]
            assert not op.attrs.get("trans_x") and not op.attrs.get("trans_y")
            assert wd[0] == chans[tid[x]], (wd, chans[tid[x]])
            emit("linear", op.out("Out"), wd[1], i=tid[x], cin=wd[0], cout=wd[1], w=w)
            i += 1
            continue
        if t == "batch_norm":
            x = op.inp("X")
            st = "bn:%s,%s,%s,%s,%r" % (op.inp("Scale"), op.inp("Bias"), op.inp("Mean"), op.inp("Variance"),
                                        float(op.attrs["epsilon"]))
            stage_on(x, op.out("Y"), st)
            i += 1
            continue
        if t in ACTS:
            stage_on(op.inp("X"), op.out("Out"), "act:" + ACTS[t])
            i += 1
            continue
        if t == "hard_sigmoid":
            stage_on(op.inp("X"), op.out("Out"),
                     "act:hsig,%r,%r" % (op_attr_f(op, "slope"), op_attr_f(op, "offset")))
            i += 1
            continue
        if t == "elementwise_add":
            x, y = op.inp("X"), op.inp("Y")
            o = op.out("Out")
            if pers(y):
                yd = V[y]["dims"]
                if yd == [1]:
                    stage_on(x, o, "sadd:" + y)
                else:
                    assert len(yd) == 1 and yd[0] == chans[tid[x]], (yd, chans[tid[x]])
                    stage_on(x, o, "bias:" + y)
            else:
                # activation + activation (possibly an upsampled view)
                if y in upsampled:
                    src, s = upsampled[y]
                    stage_on(x, o, "addup:%d,%d" % (tid[src], s))
                elif x in upsampled:
                    src, s = upsampled[x]
                    stage_on(y, o, "addup:%d,%d" % (tid[src], s))
                elif fusable(y):
                    stage_on(y, o, "addt:%d" % tid[x])
                else:
                    stage_on(x, o, "addt:%d" % tid[y])
            i += 1
            continue
        if t == "elementwise_mul":
            x, y = op.inp("X"), op.inp("Y")
            o = op.out("Out")
            if pers(x):
                assert V[x]["dims"] == [1]
                stage_on(y, o, "smul:" + x)
            else:
                # x: [N,C,H,W] activation, y: [N,C,1,1] SE gate
                assert y in producer and producer[y].kind == "sefc", (x, y)
                stage_on(x, o, "mulc:%d" % tid[y])
            i += 1
            continue
        if t == "pool2d":
            x = op.inp("X")
            c = chans[tid[x]]
            if op.attrs.get("adaptive"):
                assert op.attrs["ksize"] == [1, 1] and op.attrs["pooling_type"] == "avg"
                emit("gap", op.out("Out"), c, i=tid[x], c=c)
            else:
                assert op.attrs["paddings"] == [0, 0] and not op.attrs.get("ceil_mode")
                ks, st = op.attrs["ksize"], op.attrs["strides"]
                emit("pool", op.out("Out"), c, i=tid[x], c=c, type=op.attrs["pooling_type"],
                     kh=ks[0], kw=ks[1], sh=st[0], sw=st[1])
            i += 1
            continue
        if t == "nearest_interp_v2":
            sc = op.attrs["scale"]
            assert sc[0] == sc[1] and float(sc[0]).is_integer()
            assert not op.attrs.get("align_corners")
            upsampled[op.out("Out")] = (op.inp("X"), int(sc[0]))
            i += 1
            continue
        if t == "concat":
            assert op.attrs["axis"] == 1
            ids, ups, c = [], [], 0
            for a in op.inputs["X"]:
                if a in upsampled:
                    src, s = upsampled[a]
                    ids.append(tid[src])
                    ups.append(s)
                    c += chans[tid[src]]
                else:
                    ids.append(tid[a])
                    ups.append(1)
                    c += chans[tid[a]]
            emit("concat", op.out("Out"), c, i=ids, up=ups, c=c)
            i += 1
            continue
        if t == "layer_norm":
            x = op.inp("X")
            c = chans[tid[x]]
            emit("ln", op.out("Y"), c, i=tid[x], c=c, eps=float(op.attrs["epsilon"]),
                 g=op.inp("Scale"), b=op.inp("Bias"))
            i += 1
            continue
        if t == "softmax":
            x = op.inp("X")
            c = chans[tid[x]]
            emit("softmax", op.out("Out"), c, i=tid[x], c=c)
            i += 1
            continue
        if t == "reshape2" and list(op.attrs.get("shape", []))[2:] == [3, 8, 15]:
            # multi-head self-attention block: qkv [N,T,360] -> [N,T,120]
            x = op.inp("X")
            j = i + 1
            scale = None
            while not (ops[j].type == "reshape2" and list(ops[j].attrs.get("shape", []))[-1] == 120
                       and len(ops[j].attrs["shape"]) == 3):
                if ops[j].type == "scale":
                    scale = float(ops[j].attrs["scale"])
                    assert ops[j].attrs.get("bias", 0.0) in (0.0, None)
                j += 1
            kinds = [o.type for o in ops[i:j + 1] if o.type not in IGNORE_OPS]
            assert kinds == ["reshape2", "transpose2", "scale", "transpose2", "matmul_v2", "softmax", "dropout",
                             "matmul_v2", "transpose2", "reshape2"], kinds
            emit("attn", ops[j].out("Out"), 120, i=tid[x], heads=8, hd=15, scale=scale)
            i = j + 1
            continue
        if t in VIEW_OPS:
            x = op.inp("X")
            o = op.out("Out")
            # In NHWC with H==1 (rec neck) / 1x1 spatial (cls head) these are pure views.
            tid[o] = tid[x]
            if x in producer:
                producer[o] = producer[x]
                if producer[x].out_var == x:
                    producer[x].out_var = o
            uses[o] = uses.get(o, 0)
            i += 1
            continue
        raise NotImplementedError("op #%d %s" % (op.idx, t))
    return plan, chans


def op_attr_f(op, k):
    return float(op.attrs[k])


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outdir = os.path.join(root, "cpp-paddle-ocr_amd", "plans")
    os.makedirs(outdir, exist_ok=True)
    for name in ("det", "rec", "cls"):
        prog = Program(os.path.join(root, "models", name, "inference.pdmodel"))
        plan, chans = lower(prog)
        path = os.path.join(outdir, name + ".plan")
        with open(path, "w") as f:
            f.write("# generated by tools/make_plan.py from models/%s/inference.pdmodel - do not edit\n" % name)
            # graph signature (pd_format.cpp pdmodel_graph_signature): op count and FNV-1a 64 of the type names joined by ';'
            sig = 1469598103934665603
            for b in ";".join(o.type for o in prog.ops).encode():
                sig = ((sig ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            f.write("plan %s ntensors=%d graph_ops=%d graph_fnv=%016x\n" % (name, len(chans), len(prog.ops), sig))
            for p in plan:
                f.write(p.line() + "\n")
        kinds = {}
        for p in plan:
            kinds[p.kind] = kinds.get(p.kind, 0) + 1
        print(name, len(plan), "plan ops", kinds)


if __name__ == "__main__":
    main()
