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
ap.add_argument("--dump-graphs", default=None, help="directory for hipGraphDebugDotPrint dumps of the bottom-backward graphs")
ap.add_argument("--alias-scan", action="store_true",
                help="allocator history: for every saved region of the text encoder, list the allocations that overlap it "
                     "between its own allocation and its own free (= the allocator handed out live memory)")
ap.add_argument("--watch", action="store_true",
                help="snapshot one victim (the bf16 input the last text layer's QKV projection saved) after EVERY backward "
                     "node of the bottom segment: the first snapshot that differs names the node whose kernels wrote it")
ap.add_argument("--single-thread-backward", action="store_true",
                help="torch.autograd.set_multithreading_enabled(False): every backward node runs on the calling thread")
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
WATCH = {"victim": None, "snaps": [], "nodes": []}


def install_watch(step):
    """Post-hooks on every node below the boundary; each clones the victim (a captured copy kernel)."""
    seen, stack = set(), [t.grad_fn for t in (getattr(step.model, "_stage_boundary", []) or []) if t.grad_fn is not None]
    nodes = []
    while stack:
        fn = stack.pop()
        if fn is None or id(fn) in seen:
            continue
        seen.add(id(fn))
        nodes.append(fn)
        for nxt, _ in fn.next_functions:
            stack.append(nxt)
    WATCH["nodes"] = nodes                      # keep the wrappers (and their hooks) alive
    cand = []
    for fn in nodes:
        if type(fn).__name__ == "_LinearFnBackward":
            try:
                sv = fn.saved_tensors
            except Exception:  # noqa: BLE001
                continue
            if len(sv) >= 2 and tuple(sv[1].shape) == (2304, 768) and sv[0].shape[0] > 1000:
                cand.append((fn._sequence_nr() if hasattr(fn, "_sequence_nr") else 0, sv[0]))
    cand.sort(key=lambda c: -c[0])
    WATCH["victim"] = cand[0][1]
    print(f"[probe] watch: {len(nodes)} nodes, victim {tuple(WATCH['victim'].shape)} at {WATCH['victim'].data_ptr():#x}", flush=True)

    def mk(name):
        def hook(grad_inputs, grad_outputs):
            WATCH["snaps"].append((name, WATCH["victim"].detach().clone()))
        return hook
    for fn in nodes:
        try:
            fn.register_hook(mk(type(fn).__name__))
        except Exception:  # noqa: BLE001
            pass


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
        if args.alias_scan and HISTORY:
            report["walk_events"] = len(torch.cuda.memory._snapshot().get("device_traces", [[]])[0])
        if args.watch:
            install_watch(step)
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
st._debug_joint_bottom = not bool(args.split_bottom)
st._debug_bottom_inputs = args.bottom_inputs
st._debug_eager_g2b = bool(args.eager_g2b)
st._debug_clone_roots = bool(args.clone_roots)
st._debug_dump_graphs = args.dump_graphs
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
if args.single_thread_backward:
    torch.autograd.set_multithreading_enabled(False)
for b in batches:
    if args.taps:
        _debug.reset()
    total, _ = st.step(dict(b))
torch.cuda.synchronize()
print("loss", total.item(), "graph", st._graph is not None)
if args.watch and WATCH["snaps"]:
    ref = WATCH["snaps"][0][1].float()
    nv = 1035 if ref.shape[0] == 1400 else ref.shape[0]
    prev = None
    print(f"[probe] watch: {len(WATCH['snaps'])} snapshots")
    for i, (name, snap) in enumerate(WATCH["snaps"]):
        x = snap.float()[:nv]
        sig = (float(x.norm().item()), int((~torch.isfinite(x)).sum().item()))
        if sig != prev:
            print(f"   #{i:4d} after {name:34s} victim norm {sig[0]:.6e} nonfinite {sig[1]}")
        prev = sig
nan_names = [n for n, p in st.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(f"[probe] parameters with a non-finite gradient: {len(nan_names)}", nan_names[:6], "..." if len(nan_names) > 6 else "")
report["nan_grads"] = nan_names
if args.taps:
    # per-tap statistics over the rows that carry work (dead rows of the variable-length text path hold garbage by design)
    from sceneverse_amd import _debug
    T = _debug.TAPS
    stats = []
    n_valid_txt = None
    for k, v in T.items():
        if k.startswith("vattn.cu#"):
            n_valid_txt = int(v[-1].item())
    for k, v in T.items():
        x = v.detach().float()
        rows = None
        if k.startswith("lin") and not k.endswith(tuple(f".rows#{i}" for i in range(8))):
            tag, idx = k.split(".")[0], k.split("#")[1]
            r = T.get(f"{tag}.rows#{idx}")
            if r is not None:
                rows = int(r.item())
        elif k.startswith("vattn.") and x.dim() == 2 and n_valid_txt is not None and x.shape[0] >= n_valid_txt:
            rows = n_valid_txt
        if rows is not None and x.dim() >= 1:
            x = x[:rows]
        bad = (~torch.isfinite(x)).reshape(x.shape[0], -1) if x.dim() >= 2 else (~torch.isfinite(x)).reshape(-1, 1)
        where = None
        if bad.any():
            rr = bad.any(1).nonzero().flatten()
            cc = bad.any(0).nonzero().flatten()
            where = dict(rows=[int(rr[0]), int(rr[-1]), int(rr.numel())], cols=[int(cc[0]), int(cc[-1]), int(cc.numel())],
                         ptr=hex(v.data_ptr()))
        stats.append(dict(name=k, shape=list(v.shape), rows=rows, norm=float(x.norm().item()) if x.numel() else 0.0,
                          sum=float(x.sum().item()) if x.numel() else 0.0, nonfinite=int((~torch.isfinite(x)).sum().item()),
                          where=where))
    with open(args.taps, "w") as f:
        json.dump(stats, f, indent=0)
    print("[probe] taps:", len(T))
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
if args.alias_scan and HISTORY:
    snap = torch.cuda.memory._snapshot()
    traces = snap.get("device_traces", [[]])[0]
    print(f"[probe] alias scan over {len(traces)} allocator events, {len(regions)} regions")

    def frames_of(ev, n=6):
        out = []
        for f in ev.get("frames", []):
            fn = f.get("filename", "")
            if "sceneverse_amd" in fn or "tools/" in fn or "bench.py" in fn:
                out.append(f"{os.path.basename(fn)}:{f.get('line')} {f.get('name')}")
            if len(out) >= n:
                break
        return out
    allocs = [(k, ev) for k, ev in enumerate(traces) if ev.get("action") in ("alloc", "free_requested", "free_completed", "segment_alloc", "segment_free")]
    n_alias = 0
    for r in regions:
        if r["owner"] == "obj":
            continue
        lo, hi = r["ptr"], r["ptr"] + r["nbytes"]
        # the block that holds this region: the LAST alloc event before the walk whose range covers lo
        own = None
        for k, ev in allocs:
            if k >= report.get("walk_events", 1 << 60):
                break
            if ev["action"] == "alloc" and ev["addr"] <= lo < ev["addr"] + ev["size"]:
                own = (k, ev)
        if own is None:
            continue
        k0, ev0 = own
        freed_at = None
        for k, ev in allocs:
            if k > k0 and ev["action"] == "free_requested" and ev["addr"] == ev0["addr"]:
                freed_at = k
                break
        for k, ev in allocs:
            if k <= k0 or (freed_at is not None and k >= freed_at):
                continue
            if ev["action"] == "alloc" and ev["addr"] < hi and ev["addr"] + ev["size"] > lo:
                n_alias += 1
                print(f"  ALIAS: {r['owner']} {r['node']} {r['slot']} {r['shape']} block [{ev0['addr']:#x}, +{ev0['size']}) alloc #{k0} "
                      f"free #{freed_at}; overlapped by alloc #{k} [{ev['addr']:#x}, +{ev['size']}) {frames_of(ev)}")
        if r["node"] in ("_LinearFnBackward", "_FusedVarlenSelfAttentionBackward") and r["shape"][0] >= 1000:
            print(f"  region {r['owner']} {r['node']} {r['slot']} {r['shape']} [{lo:#x}, {hi:#x}) block alloc #{k0} by {frames_of(ev0, 3)} free #{freed_at}")
    print(f"[probe] alias scan: {n_alias} overlapping allocations of live text-encoder regions")
    # who lives right below / above the first victim (the bf16 input the last text layer's QKV projection saved) while
    # the bottom backward is captured?  An overrun of a neighbour is the other way to reach it.
    vic = [r for r in regions if r["owner"] != "obj" and r["node"] == "_LinearFnBackward" and r["slot"] == "saved[0]" and r["shape"][0] >= 1000]
    if vic:
        lo, hi = vic[0]["ptr"], vic[0]["ptr"] + vic[0]["nbytes"]
        print(f"[probe] neighbours of the victim [{lo:#x}, {hi:#x}) after the walk (event {report.get('walk_events')}):")
        for k, ev in allocs:
            if k < report.get("walk_events", 0) or ev["action"] != "alloc":
                continue
            end = ev["addr"] + ev["size"]
            if (lo - (8 << 20) <= end <= lo) or (hi <= ev["addr"] <= hi + (1 << 20)):
                print(f"    #{k} [{ev['addr']:#x}, {end:#x}) size {ev['size']} gap-to-victim {lo - end if end <= lo else ev['addr'] - hi} {frames_of(ev, 3)}")
with open(args.out, "w") as f:
    json.dump(report, f, indent=1, default=str)
print("[probe] wrote", args.out)
