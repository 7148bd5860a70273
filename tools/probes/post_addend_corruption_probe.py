"""Root-cause probe for the "post-addend" corruption of the split-graph step:

    python tools/probes/post_addend_corruption_probe.py OUT.json [--post spatial|plain|all|off] [--graph dp|one]

What it does, in one process:
  1. records the caching allocator's history (torch.cuda.memory._record_memory_history, python stacks);
  2. after graph 1 (forward) is captured, walks the autograd graph below every stage-boundary tensor and notes the
     address range of every tensor a backward node SAVED (who saved it, shape, dtype) -- without keeping references,
     so lifetimes are exactly the product's;
  3. on the first REPLAY, copies those ranges to the host after graph 1, after the eager gather and after graph 2a: a
     saved activation of the bottom segment must not change before graph 2b consumes it;
  4. for every range that did change, lists the allocator events (alloc / free, with their python frames) that touched
     the changed bytes -- i.e. who freed the activation early and who was handed its memory.
"""
import argparse
import ctypes
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--post", default="spatial")
ap.add_argument("--graph", default="dp")
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--n-obj", type=int, default=16)
ap.add_argument("--save-grads", default=None)
ap.add_argument("--split-bottom", action="store_true", help="object-encoder and text-encoder backward as separate graphs")
ap.add_argument("--no-varlen", action="store_true")
ap.add_argument("--bottom-inputs", default=None, choices=["lang_encoder", "point_encoder"])
ap.add_argument("--eager-g2b", action="store_true")
ap.add_argument("--clone-roots", action="store_true")
ap.add_argument("--capture-mode", default=None, choices=["global", "thread_local", "relaxed"])
ap.add_argument("--no-wgrad-group", action="store_true", help="classic weight gradients: one split-K GEMM + reduce per Linear")
ap.add_argument("--ws-mode", default="cached", choices=["cached", "fresh", "prealloc"],
                help="split-K workspace: the product's cached growing buffer | a fresh buffer per call | one 256 MB buffer "
                     "made before any capture")
ap.add_argument("--no-split", action="store_true", help="weight gradients without split-K (splits = 1)")
ap.add_argument("--taps", default=None, help="save the debug taps (sceneverse_amd/_debug.py) of the last step here")
ap.add_argument("--obj-first", action="store_true", help="run the object encoder BEFORE the text encoder in forward")
ap.add_argument("--no-cls-tail", action="store_true")
ap.add_argument("--fill-nan", action="store_true",
                help="every torch.empty is filled with NaN (torch.utils.deterministic.fill_uninitialized_memory; the fills "
                     "are captured into the graphs too): a kernel that reads memory nobody wrote shows up as NaN whatever "
                     "the allocator handed out")
args = ap.parse_args()

from bench import gps_pretrain_cfg, _lang_dir
from sceneverse_amd.data.synthetic import synth_batch
from sceneverse_amd.engine import GPSTrainStep
from sceneverse_amd.modules.layers import transformers as T

DEV = "cuda"
if args.fill_nan:
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
if args.post != "off" and hasattr(T, "set_fuse_post_add"):
    T.set_fuse_post_add(True, None if args.post == "all" else args.post)
try:
    torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python", max_entries=2000000)
    HISTORY = True
except Exception as e:  # noqa: BLE001
    print("memory history unavailable:", e)
    HISTORY = False

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def dtoh(ptr: int, nbytes: int) -> np.ndarray:
    buf = np.empty(nbytes, dtype=np.uint8)
    rc = hip.hipMemcpy(buf.ctypes.data, ctypes.c_void_p(ptr), nbytes, 2)
    assert rc == 0, rc
    return buf


regions = []          # dict(owner, node, slot, ptr, nbytes, shape, dtype)
snaps = {}            # stage -> [np.ndarray per region]
report = {"post": args.post, "graph": args.graph, "changed": [], "history": HISTORY}


KEEP = []     # python wrappers of the visited nodes stay alive during the walk: a recycled id() would truncate it


def walk(owner: str, t: torch.Tensor, seen_nodes: set, seen_ptr: set):
    stack = [t.grad_fn] if t.grad_fn is not None else []
    while stack:
        fn = stack.pop()
        if fn is None or id(fn) in seen_nodes:
            continue
        seen_nodes.add(id(fn))
        KEEP.append(fn)
        saved = []
        try:
            if hasattr(fn, "saved_tensors"):
                saved = [(f"saved[{i}]", s) for i, s in enumerate(fn.saved_tensors)]
        except Exception:  # noqa: BLE001
            saved = []
        for name in dir(fn):
            if name.startswith("_saved_"):
                try:
                    v = getattr(fn, name)
                except Exception:  # noqa: BLE001
                    continue
                if torch.is_tensor(v):
                    saved.append((name, v))
                elif isinstance(v, (list, tuple)):
                    saved += [(f"{name}[{i}]", u) for i, u in enumerate(v) if torch.is_tensor(u)]
        # tensors python Functions keep as plain ctx attributes
        for name in ("rows_dev",):
            v = getattr(fn, name, None)
            if torch.is_tensor(v):
                saved.append((f"ctx.{name}", v))
        for slot, s in saved:
            if s is None or not s.is_cuda or s.numel() == 0:
                continue
            nbytes = s.untyped_storage().nbytes() - s.storage_offset() * s.element_size()
            nb = min(nbytes, s.numel() * s.element_size()) if s.is_contiguous() else nbytes
            key = (s.data_ptr(), nb)
            if key in seen_ptr:
                continue
            seen_ptr.add(key)
            regions.append(dict(owner=owner, node=type(fn).__name__, slot=slot, ptr=s.data_ptr(), nbytes=int(nb),
                                shape=list(s.shape), dtype=str(s.dtype)))
        del saved
        for nxt, _ in fn.next_functions:
            stack.append(nxt)


first_replay = {"done": False}


def hook(stage, step, **kw):
    if stage == "captured_g1":
        seen_nodes, seen_ptr = set(), set()
        names = ["txt", "scene_txt", "obj"]
        for i, t in enumerate(getattr(step.model, "_stage_boundary", []) or []):
            walk(names[i] if i < len(names) else f"b{i}", t, seen_nodes, seen_ptr)
        # parameters and the static batch are not activations
        skip = {p.data_ptr() for p in step.model.parameters()} | {b.data_ptr() for b in step.model.buffers()}
        regions[:] = [r for r in regions if r["ptr"] not in skip]
        KEEP.clear()
        print(f"[probe] {len(regions)} saved regions below the boundary, {sum(r['nbytes'] for r in regions) / 1e6:.1f} MB")
    elif stage.startswith("replayed_") and not first_replay["done"]:
        torch.cuda.synchronize()
        print("[probe] stage ok:", stage, flush=True)
        snaps[stage] = [dtoh(r["ptr"], r["nbytes"]) for r in regions]
        if stage == "replayed_g2b":
            first_replay["done"] = True


if args.taps:
    from sceneverse_amd import _debug
    _debug.ENABLED = True
if args.obj_first:
    import sceneverse_amd.model.openvocab as OV
    OV._OBJ_FIRST = True
from sceneverse_amd.modules.layers import gemm as _G
if args.ws_mode == "fresh":
    _G._workspace = lambda device, floats: torch.empty(max(floats, 1), dtype=torch.float32, device=device)
elif args.ws_mode == "prealloc":
    _BIG = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    _G._workspace = lambda device, floats: _BIG
if args.no_split:
    _lib = _G._native.load()
    _orig_pick = _lib.gps_gemm_pick_splits

    class _NoSplitLib:
        def __getattr__(self, name):
            if name == "gps_gemm_pick_splits":
                return lambda *a: 1
            return getattr(_lib, name)
    _G._native.load = lambda *a, **k: _NoSplitLib()
cfg = gps_pretrain_cfg(_lang_dir())
st = GPSTrainStep(cfg, device=DEV, ddp=False, graph={"dp": "dp", "one": True, "off": False}[args.graph], graph_warmup=2, seed=7,
                  wgrad_group=not args.no_wgrad_group)
st.stage_hook = hook
st._debug_split_bottom = bool(args.split_bottom)
st._debug_bottom_inputs = args.bottom_inputs
st._debug_eager_g2b = bool(args.eager_g2b)
st._debug_clone_roots = bool(args.clone_roots)
if args.capture_mode:
    import sceneverse_amd.engine as _E
    _E._CAPTURE_MODE = args.capture_mode
if args.no_varlen or args.no_cls_tail:
    from sceneverse_amd.modules.language import bert as _B
    if args.no_varlen:
        _B.set_varlen(False)
    if args.no_cls_tail:
        _B.set_cls_tail(False)
for m in st.model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if isinstance(m, T.MultiheadSelfAttention):
        m.dropout = 0.0
    if hasattr(m, "attention_probs_dropout_prob"):
        m.attention_probs_dropout_prob = 0.0
junk = [torch.full((256, 1024, 1024), float("nan"), device=DEV) for _ in range(20)]
del junk
batches = [synth_batch(args.batch, n_obj=args.n_obj, seed=20 + i, min_real=5, device=DEV) for i in range(3)]
for b in batches:
    if args.taps:
        _debug.reset()
    total, _ = st.step(dict(b))
torch.cuda.synchronize()
print("loss", total.item(), "graph", st._graph is not None)
nan_names = [n for n, p in st.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(f"[probe] parameters with a non-finite gradient: {len(nan_names)}", nan_names[:6], "..." if len(nan_names) > 6 else "")
report["nan_grads"] = nan_names
if args.taps:
    from sceneverse_amd import _debug
    torch.save({k: v.detach().float().cpu() for k, v in _debug.TAPS.items()}, args.taps)
    print("[probe] taps:", len(_debug.TAPS))
if args.save_grads:
    torch.save({n: p.grad.detach().to(torch.bfloat16).cpu() for n, p in st.model.named_parameters() if p.grad is not None},
               args.save_grads)

# ---- which saved regions changed before graph 2b read them? ------------------------------------------------
order = [s for s in ("replayed_g1", "replayed_gather", "replayed_g2a", "replayed_g2b_part0") if s in snaps]
changed = []
for i, r in enumerate(regions if order else []):
    base = snaps[order[0]][i]
    for later in order[1:]:
        cur = snaps[later][i]
        if not np.array_equal(base, cur):
            d = np.nonzero(base != cur)[0]
            w = slice(int(d[0]) // 4 * 4, int(d[0]) // 4 * 4 + 32)
            changed.append(dict(r, stage=later, first=int(d[0]), last=int(d[-1]), n_diff=int(d.size),
                                before_i32=base[w].view(np.int32).tolist(), after_i32=cur[w].view(np.int32).tolist(),
                                before_f32=base[w].view(np.float32).tolist(), after_f32=cur[w].view(np.float32).tolist(),
                                crc_before=zlib.crc32(base.tobytes()), crc_after=zlib.crc32(cur.tobytes())))
            break
print(f"[probe] {len(changed)} of {len(regions)} saved regions CHANGED between graph 1 and graph 2b")
for c in changed[:40]:
    print(f"  {c['owner']:9s} {c['node']:28s} {c['slot']:14s} {c['shape']} {c['dtype']} ptr {c['ptr']:#x} +[{c['first']}, {c['last']}] "
          f"({c['n_diff']} bytes differ) after {c['stage']}\n      i32 {c['before_i32']} -> {c['after_i32']}\n      f32 {c['before_f32']} -> {c['after_f32']}")
report["regions"] = len(regions)
report["changed"] = changed

if HISTORY and changed:
    snap = torch.cuda.memory._snapshot()
    traces = snap.get("device_traces", [[]])[0]
    print(f"[probe] allocator history: {len(traces)} events")

    def frames_of(ev, n=7):
        out = []
        for f in ev.get("frames", []):
            fn = f.get("filename", "")
            if "sceneverse_amd" in fn or "tools/" in fn or "bench.py" in fn:
                out.append(f"{os.path.relpath(fn, os.getcwd()) if fn.startswith('/') else fn}:{f.get('line')} {f.get('name')}")
            if len(out) >= n:
                break
        return out

    detail = []
    for c in changed[:6]:
        lo, hi = c["ptr"] + c["first"], c["ptr"] + c["last"] + 1
        evs = []
        for k, ev in enumerate(traces):
            a, sz = ev.get("addr"), ev.get("size")
            if a is None or sz is None or ev.get("action") not in ("alloc", "free_requested", "free_completed", "free"):
                continue
            if a < hi and a + sz > lo:
                evs.append(dict(k=k, action=ev["action"], addr=a, size=sz, stream=ev.get("stream"), frames=frames_of(ev)))
        print(f"--- events touching {c['owner']} {c['node']} {c['slot']} [{lo:#x}, {hi:#x}): {len(evs)}")
        for e in evs[-14:]:
            print(f"   #{e['k']} {e['action']:15s} addr {e['addr']:#x} size {e['size']} stream {e['stream']}")
            for fr in e["frames"]:
                print("        ", fr)
        detail.append(dict(region={k: c[k] for k in ("owner", "node", "slot", "shape", "dtype", "ptr", "first", "last")}, events=evs[-40:]))
    report["events"] = detail
with open(args.out, "w") as f:
    json.dump(report, f, indent=1, default=str)
print("[probe] wrote", args.out)
