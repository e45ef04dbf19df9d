"""torch.autograd bindings of the HIP hot path (include/hoisdf.h via hoisdf_amd._lib).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape.  All
arithmetic of the hot path runs in libhoisdf_hip.so.  There is no CPU or eager fallback: a
CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import torch

from ._lib import Pyramid, call, lib

_SEED = [0x9E3779B97F4A7C15, 0]
_WEIGHT_GEN = [0]      # bumped by optimizers that update parameters behind autograd's back (FusedAdamW's HIP kernel does
                       # not touch torch's version counters): invalidates cached derived weights (SdfQueryWeights)


def deterministic() -> bool:
    """HOISDF_DETERMINISTIC=1 / ops.set_deterministic(True): order-fixed forms of every accumulating kernel (see
    include/hoisdf.h hoisdf_set_deterministic); the model then also runs single-stream."""
    from ._lib import lib
    return bool(lib().hoisdf_get_deterministic())


def set_deterministic(on: bool) -> None:
    from ._lib import lib
    lib().hoisdf_set_deterministic(int(bool(on)))


def bump_weight_generation() -> None:
    _WEIGHT_GEN[0] += 1
    _emu_refresh_images()


def manual_seed(seed: int) -> None:
    """Seed of the counter-based dropout RNG (each dropout site draws one 64-bit stream id)."""
    _SEED[0] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _SEED[1] = 0


def next_seed() -> int:
    _SEED[1] += 1
    return (_SEED[0] * 6364136223846793005 + _SEED[1] * 1442695040888963407) & 0xFFFFFFFFFFFFFFFF


class _ZeroArena:
    """Per-step arena of zero-initialised gradient scratch (weight / bias gradients the split-k GEMM accumulates into
    with atomics, LayerNorm gamma/beta sums, ...): ONE memset per step instead of one fill kernel per backward op
    (~170 launches per training step).  Only active between ``zero_arena_begin_step()`` and ``zero_arena_end_step()``
    (GradReducer.zero_grad() ... GradReducer.finish()): the views handed out are only valid until the next ``begin_step`` -
    fine for gradients that are packed into the reducer's buckets before the optimizer reads them; any backward that runs
    outside that window (another model, a test, a gradient kept across steps) gets plain ``torch.zeros``.  The first step
    measures the demand, the buffer is sized from it."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.demand = 0
        self.active = False
        self.measured = False

    def begin_step(self, device):
        if self.measured and self.buf is None and self.demand > 0:
            self.buf = torch.empty(int(self.demand * 1.25) + 1024, device=device, dtype=torch.float32)
        self.active = True
        self.measured = True
        if self.buf is not None:
            (self.buf[:self.off] if 0 < self.off < self.buf.numel() else self.buf).zero_()   # only what the last step used
        self.off = 0
        self.demand = 0

    def zeros(self, n, device):
        n_al = (n + 15) // 16 * 16                 # 64-byte aligned slices
        if self.active:
            self.demand += n_al
        if not self.active:
            return torch.zeros(n, device=device, dtype=torch.float32)
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            return torch.zeros(n, device=device, dtype=torch.float32)
        v = self.buf[self.off:self.off + n]
        self.off += n_al
        return v


_ARENA = _ZeroArena()


def zero_arena_begin_step(device) -> None:
    """Call at the start of every training step (before forward) to serve the backward's zero-initialised scratch from one
    buffer that is cleared with a single memset.  See _ZeroArena for the validity contract."""
    _ARENA.begin_step(torch.device(device))
    _EMU_PLANES.clear()            # kept attention planes of graphs that never ran their backward


def zero_arena_end_step() -> None:
    """the step's gradients have been packed / consumed: backward passes from here on get plain torch.zeros again"""
    _ARENA.active = False


zero_arena_disable = zero_arena_end_step


def _zeros(n: int, device) -> torch.Tensor:
    return _ARENA.zeros(int(n), torch.device(device) if not isinstance(device, torch.device) else device)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype not in (torch.float32,):
            raise RuntimeError("hoisdf_amd ops need float32 CUDA/HIP tensors (no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:       # the C ABI launches on the calling thread's current device and stream
            raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: one process per "
                               "GPU, call torch.cuda.set_device(local_rank) first")


def _rows(x: torch.Tensor) -> torch.Tensor:
    """view as (M, K) with unit inner stride and a uniform row stride"""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2


# ---------------------------------------------------------------------------------------------
# pyramid handling
# ---------------------------------------------------------------------------------------------
_SHARED_PYR_GRAD = __import__("os").environ.get("HOISDF_SHARED_PYR_GRAD", "1") != "0"      # A/B switch


class PyramidNHWC:
    """The feature pyramid in the layout the kernels want: per level a contiguous
    [B][H][W][C] float32 tensor (a zero-copy view when the encoder ran channels_last)."""

    def __init__(self, levels: Sequence[torch.Tensor]):
        self.levels = [l if l.is_contiguous() else l.contiguous() for l in levels]
        _chk(*self.levels)
        self.B = self.levels[0].shape[0]
        self.C = sum(l.shape[3] for l in self.levels)

    @staticmethod
    def from_nchw(maps: Sequence[torch.Tensor]) -> "PyramidNHWC":
        return PyramidNHWC([m.permute(0, 2, 3, 1) for m in maps])

    acc = None       # set by shared_grad(): the accumulator every project_gather backward of this pyramid scatters into

    def shared_grad(self) -> "PyramidNHWC":
        """Training: the pyramid is gathered from several times per step (hand points, object points, the two SDF-loss
        point sets) and every gather's backward used to fill its own five zeroed level gradients, which autograd then
        summed (15 adds over 3 x 130 MB + 4 x 130 MB of fills at B = 32).  The returned pyramid routes all of them into ONE set
        of zeroed buffers (float atomics already accumulate) that a sink node hands to the encoder once, after the last
        gather backward.  No-op without gradients and in deterministic mode (the order-fixed gather owns its output)."""
        if not (torch.is_grad_enabled() and any(l.requires_grad for l in self.levels)) or deterministic() or not _SHARED_PYR_GRAD:
            return self
        acc = _PyrAcc([tuple(l.shape) for l in self.levels], self.levels[0].device)
        out = PyramidNHWC(list(_PyrSink.apply(acc, *self.levels)))
        out.acc = acc
        return out

    def struct(self, tensors: Optional[Sequence[torch.Tensor]] = None) -> Pyramid:
        ts = self.levels if tensors is None else tensors
        s = Pyramid()
        s.n_levels, s.B = len(ts), self.B
        for i, t in enumerate(ts):
            s.data[i] = t.data_ptr()
            s.C[i], s.H[i], s.W[i] = t.shape[3], t.shape[1], t.shape[2]
        return s


class _PyrAcc:
    """shared level-gradient buffers of one step (PyramidNHWC.shared_grad): zeroed at creation (a slice of the per-step
    arena when it is active), written by every gather backward on whatever stream that node runs on"""

    def __init__(self, shapes, device):
        self.shapes = shapes
        sizes = [math.prod(sh) for sh in shapes]
        flat = _zeros(sum(sizes), device)
        self.bufs, off = [], 0
        for sh, n in zip(shapes, sizes):
            self.bufs.append(flat[off:off + n].view(sh))
            off += n
        self.flat = flat
        self.events = []          # one per gather backward: the sink waits for them (other streams included)
        self.touched = False


class _PyrSink(torch.autograd.Function):
    """identity on the pyramid levels; its backward runs after every consumer's and returns the accumulated gradients"""

    @staticmethod
    def forward(ctx, acc, *levels):
        ctx.acc = acc
        ctx.set_materialize_grads(False)
        return tuple(l.view_as(l) for l in levels)

    @staticmethod
    def backward(ctx, *gs):
        acc = ctx.acc
        cur = torch.cuda.current_stream(acc.flat.device)
        for ev in acc.events:
            cur.wait_event(ev)
        acc.flat.record_stream(cur)
        out = []
        for g, b in zip(gs, acc.bufs):
            if g is not None:                 # a consumer other than project_gather returned a real gradient
                b = b + g if acc.touched else g
            elif not acc.touched:
                b = None
            out.append(b)
        return (None, *out)


class _ProjectGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, sample_idx, center, cam_intr, scale, img_hw, acc, *levels):
        ctx.acc = acc
        pyr = PyramidNHWC(levels)
        pts = points.reshape(-1, 3).contiguous()
        _chk(pts, center, cam_intr)
        n = pts.shape[0]
        rps = points.shape[-2] if sample_idx is None else 1
        feat = torch.empty(n, pyr.C, device=pts.device, dtype=torch.float32)
        cam = torch.empty(n, 3, device=pts.device, dtype=torch.float32)
        s = pyr.struct()
        call("hoisdf_project_gather_fwd", C.byref(s), _p(pts), _p(sample_idx), n, rps, _p(center), _p(cam_intr),
             float(scale), img_hw[0], img_hw[1], _p(feat), pyr.C, _p(cam), None, _st())
        ctx.save_for_backward(pts, sample_idx, center, cam_intr)
        ctx.meta = (float(scale), img_hw, rps, [tuple(l.shape) for l in pyr.levels], pyr.B)
        ctx.mark_non_differentiable(cam)
        return feat, cam

    @staticmethod
    def backward(ctx, dfeat, _dcam):
        pts, sample_idx, center, cam_intr = ctx.saved_tensors
        scale, img_hw, rps, shapes, B = ctx.meta
        dfeat = dfeat.contiguous()
        acc = ctx.acc
        if acc is not None:
            grads = acc.bufs
            acc.flat.record_stream(torch.cuda.current_stream(dfeat.device))
        else:
            grads = [torch.zeros(sh, device=dfeat.device, dtype=torch.float32) for sh in shapes]
        g = Pyramid()
        g.n_levels, g.B = len(grads), B
        for i, t in enumerate(grads):
            g.data[i] = t.data_ptr()
            g.C[i], g.H[i], g.W[i] = t.shape[3], t.shape[1], t.shape[2]
        call("hoisdf_project_gather_bwd", C.byref(g), _p(pts), _p(sample_idx), pts.shape[0], rps, _p(center),
             _p(cam_intr), scale, img_hw[0], img_hw[1], _p(dfeat), dfeat.shape[1], _st())
        if acc is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dfeat.device))
            acc.events.append(ev)
            acc.touched = True
            return (None,) * (7 + len(shapes))
        return (None, None, None, None, None, None, None, *grads)


def project_gather(pyr: PyramidNHWC, points, center, cam_intr, scale, img_hw=(256, 256), sample_idx=None):
    """K1. points (B,P,3) [or (n,3) with sample_idx] -> feat (n, C), cam (n, 3).
    Differentiable w.r.t. the pyramid levels only (the grid is detached in the reference)."""
    return _ProjectGather.apply(points, sample_idx, center.contiguous(), cam_intr.contiguous(), scale,
                                tuple(img_hw), pyr.acc, *pyr.levels)


# ---------------------------------------------------------------------------------------------
# linear
# ---------------------------------------------------------------------------------------------
# ---- fp32 emulated on the bf16 MFMA pipe (csrc/gemm_emu.hip): forward / grad-input of the large linear layers ------------
_GEMM_EMU = __import__("os").environ.get("HOISDF_GEMM", "emu") != "f32"
_GEMM_EMU_MIN_ROWS = 2048            # below this a problem is a handful of tiles: latency-bound, stays on the f32 kernel
_GEMM_EMU_DW_MIN_ROWS = 8192         # grad-weight: the contraction runs over the rows (>= 32 slabs per slice at 256 slices)
_GEMM_EMU_DW_MIN_WIDTH = int(__import__("os").environ.get("HOISDF_EMU_DW_MIN_WIDTH", "64"))
_EMU_IMAGES = {}                     # (device, data_ptr, shape, ld, transpose) -> [image, version key, event, build stream, owner, reader streams]
_EMU_PURGE_AT = [4096]


def set_gemm_emu(on: bool) -> None:
    """cfg.gemm_emu (default on; HOISDF_GEMM=f32 turns it off): forward and grad-input of the large linear layers as exact
    three-way bf16 splits of both f32 operands, six bf16 MFMA products per product, f32 accumulation - fp32-equivalent
    results (include/hoisdf.h hoisdf_linear_fwd_emu) at 1.7x the f32 MFMA kernel.  Off: the exact-f32 MFMA kernel."""
    global _GEMM_EMU
    _GEMM_EMU = bool(on)
    from ._lib import lib
    lib().hoisdf_set_gemm_emu(int(_GEMM_EMU))            # the layers inside hoisdf_sdf_query_fwd follow


def gemm_emu() -> bool:
    return _GEMM_EMU


_EMU_SMALL = __import__("os").environ.get("HOISDF_EMU_SMALL", "1") != "0"
_EMU_SMALL_MAX = [None]


def _emu_small_max_rows() -> int:
    if _EMU_SMALL_MAX[0] is None:
        from ._lib import lib
        _EMU_SMALL_MAX[0] = lib().hoisdf_linear_emu_small_max_rows()
    return _EMU_SMALL_MAX[0]


def _emu_small_ok(M: int, a: torch.Tensor, lda: int, W: torch.Tensor, N: int, K: int) -> bool:
    """small row counts (the 17-query decoder stack, the heads): the one-wave-per-tile emulated kernels (hoisdf_linear_*_emu_small)"""
    if not (_GEMM_EMU and _EMU_SMALL):
        return False
    return (1 <= M <= _emu_small_max_rows() and N % 4 == 0 and K % 4 == 0 and lda % 4 == 0 and W.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0)


def _emu_ok(M: int, a: torch.Tensor, lda: int, contraction: int) -> bool:
    return (_GEMM_EMU and M >= _GEMM_EMU_MIN_ROWS and contraction % 4 == 0 and lda % 4 == 0
            and a.data_ptr() % 16 == 0)


_EMU_GRAVEYARD = []                  # images of weights that no longer exist, kept for one more purge cycle (see _emu_purge)


def _emu_purge() -> None:
    """Drop the cache entries whose weight tensor is gone.  Never touches an entry whose owner is alive: callers (the coarse
    encoder / decoder layer entries) hold only the raw device pointer of an image for the duration of their C call, so an
    image must not be freed behind a live weight.  The purged images themselves are parked until the NEXT purge (thousands
    of lookups later), far beyond any kernel that may still read them on another stream."""
    _EMU_GRAVEYARD.clear()
    for k in [k for k, e in _EMU_IMAGES.items() if e[4]() is None]:
        _EMU_GRAVEYARD.append(_EMU_IMAGES.pop(k)[0])
    _EMU_EPOCH[0] += 1


_EMU_EPOCH = [0]                     # bumped whenever an entry joins or leaves _EMU_IMAGES: the batch table is rebuilt
_EMU_TABLE = {}                      # device index -> (epoch, keys, device table, n, total_blocks)
_EMU_BATCH = __import__("os").environ.get("HOISDF_EMU_BATCH_PREP", "1") != "0"


def _emu_refresh_images() -> None:
    """After an optimizer step (bump_weight_generation): rebuild EVERY cached weight image whose parameter is alive in one launch
    per device (hoisdf_linear_emu_prepare_batch) on the current stream, instead of ~170 few-microsecond launches strewn over the
    next step's critical path.  An entry keeps its device pointer (data_ptr is in the key), so the device-side table is built
    once per cache composition.  HOISDF_EMU_BATCH_PREP=0: images are rebuilt one by one at their first use (the round-3 flow)."""
    if not _EMU_BATCH or not _EMU_IMAGES:
        return
    from ._lib import EmuPrepItem, lib
    import ctypes as C
    by_dev = {}
    for k, e in list(_EMU_IMAGES.items()):
        base = e[4]()
        if base is None:
            continue
        # the owner must still COVER the cached address: param.data = ..., module.to() / .float(), load_state_dict(assign=True) keep the
        # Parameter object alive but free its old storage - the batch kernel must never read that
        try:
            st = base.untyped_storage()
            lo = st.data_ptr()
            ok = base.is_cuda and base.device.index == k[0] and lo <= k[1] and k[1] + 4 * ((k[2] - 1) * k[4] + k[3]) <= lo + st.nbytes()
        except RuntimeError:
            ok = False
        if not ok:
            _EMU_GRAVEYARD.append(_EMU_IMAGES.pop(k)[0])
            _EMU_EPOCH[0] += 1
            continue
        by_dev.setdefault(k[0], []).append((k, e))
    for dev, ents in by_dev.items():
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            tab = _EMU_TABLE.get(dev)
            if tab is None or tab[0] != _EMU_EPOCH[0] or tab[1] != [k for k, _ in ents]:     # (a weight died: never read freed memory)
                arr = (EmuPrepItem * len(ents))()
                blk = 0
                for i, (k, e) in enumerate(ents):
                    _, ptr, N, K, ldw, tr = k
                    arr[i].W, arr[i].image, arr[i].first_block = ptr, e[0].data_ptr(), blk
                    arr[i].ldw, arr[i].N, arr[i].K, arr[i].transpose = ldw, N, K, int(tr)
                    blk += lib().hoisdf_linear_emu_prepare_blocks(N, K, int(tr))
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                tab = (_EMU_EPOCH[0], [k for k, _ in ents], host.to(f"cuda:{dev}"), len(ents), blk)
                _EMU_TABLE[dev] = tab
            readers = set()
            for _, e in ents:
                readers |= e[5]
            for s_ in readers:
                if s_ != cur:
                    cur.wait_stream(s_)      # every stream that read a previous image
            call("hoisdf_linear_emu_prepare_batch", _p(tab[2]), tab[3], tab[4], _st())
            ev = torch.cuda.Event()
            ev.record(cur)
            for _, e in ents:
                base = e[4]()
                if base is None:
                    continue
                e[5].clear()
                e[5].add(cur)                # the building stream: a rebuild from another stream has to wait for this build too
                e[1], e[2], e[3] = (_WEIGHT_GEN[0], base._version), ev, cur


def _emu_image(W: torch.Tensor, transpose: bool) -> torch.Tensor:
    """the bf16x3 slab image of a weight (hoisdf_linear_emu_prepare), cached per (device, storage, shape, orientation) and
    rebuilt in place when the weight changed (torch's version counter, or the generation FusedAdamW bumps).  The build is
    recorded with an event: a consumer on another HIP stream (the object stack runs on a second one) waits for it, and is
    remembered as a reader - a rebuild waits for every stream that read the previous image before overwriting it."""
    from ._lib import lib
    N, K = W.shape
    key = (W.device.index, W.data_ptr(), N, K, W.stride(0), bool(transpose))
    ver = (_WEIGHT_GEN[0], W._version)
    cur = torch.cuda.current_stream(W.device)
    ent = _EMU_IMAGES.get(key)
    # the entry belongs to ONE tensor object (the parameter, or the parameter a slice views): another tensor that the
    # allocator later placed at the same address must not hit it
    base = W._base if W._base is not None else W
    if ent is not None and ent[4]() is not base:
        ent[1], ent[4] = None, __import__("weakref").ref(base)
    if ent is None:
        nb = lib().hoisdf_linear_emu_image_bytes(K if transpose else N, N if transpose else K)
        ent = [torch.empty(nb, device=W.device, dtype=torch.uint8), None, None, None, __import__("weakref").ref(base), set()]
        if len(_EMU_IMAGES) >= _EMU_PURGE_AT[0]:     # weights that came and went (tests): do not grow without bound
            _emu_purge()
            _EMU_PURGE_AT[0] = max(4096, 2 * len(_EMU_IMAGES))
        _EMU_IMAGES[key] = ent
        _EMU_EPOCH[0] += 1
    if ent[1] != ver:
        for s_ in ent[5]:
            if s_ != cur:
                cur.wait_stream(s_)          # every stream that read the previous image (the builder included)
        ent[5].clear()
        call("hoisdf_linear_emu_prepare", _p(W), W.stride(0), N, K, int(transpose), _p(ent[0]), _st())
        ev = torch.cuda.Event()
        ev.record(cur)
        ent[1], ent[2], ent[3] = ver, ev, cur
    elif ent[3] != cur:
        cur.wait_event(ent[2])
    ent[5].add(cur)
    return ent[0]


_H2 = [None]


def _h2() -> bool:
    """the process runs the f16x2 form of the emulated contractions (include/hoisdf.h, HOISDF_EMU_FORM)"""
    if _H2[0] is None:
        from ._lib import lib
        _H2[0] = lib().hoisdf_linear_emu_pieces() == 2
    return _H2[0]


def _mag_measure(a2, lda, M, K):
    """row magnitudes (f16x2 form, include/hoisdf.h: one word per row) of an operand whose producer left none: one read of it
    (hoisdf_mag_measure), so that the contractions that consume it - forward and grad-weight of a layer, grad-input and
    grad-weight of its backward - do not each measure it again.  None outside the f16x2 form."""
    if not _h2():
        return None
    words = torch.empty(max(int(M), 1), device=a2.device, dtype=torch.int32)
    call("hoisdf_mag_measure", _p(a2), lda, M, K, _p(words), _st())
    a2._hoisdf_words = (words, a2._version, a2.data_ptr(), lda, M, K)
    return words


def _head_measure(a2, lda, M, groups, L):
    """head magnitudes (include/hoisdf.h) of an attention operand matrix: one word per (64-column group, sample of L rows) ->
    int32 tensor [groups][M / L] (the attention entries take one scale per (sample, head) from it)"""
    nb = (M + L - 1) // L
    words = torch.empty(groups * nb, device=a2.device, dtype=torch.int32)
    call("hoisdf_head_mag_measure", _p(a2), lda, M, groups, L, _p(words), _st())
    return words


def _mag_known(a2, lda, M, K):
    """the words an earlier contraction over this very tensor object measured (same storage version, same view).  Only the grad-weight
    looks here - it finds what the grad-input of the same layer just measured for dy, or the forward for x: operands nothing writes in
    between (the C entries write through raw pointers, which torch's version counter does not see: forward / grad-input always measure)"""
    t = getattr(a2, "_hoisdf_words", None)
    if t is not None and t[1:] == (a2._version, a2.data_ptr(), lda, M, K):
        return t[0]
    return None


def _gemm_fwd(x2, ldx, W, b, y, ldy, M, N, K, act, drop_p, seed, bits, x_mag=None, y_mag=None):
    """-> the row magnitudes of x2 that the emulated f16x2 contraction used (given or measured here; None otherwise);
    y_mag (zero-filled int32 [M]): receives y's row magnitudes from the epilogue (f16x2 form)"""
    if _emu_ok(M, x2, ldx, K):
        if _h2():
            if x_mag is None:
                x_mag = _mag_measure(x2, ldx, M, K)
            call("hoisdf_linear_fwd_emu_mag", _p(x2), ldx, _p(_emu_image(W, False)), _p(b), _p(y), ldy, M, N, K, int(act),
                 float(drop_p), seed, _p(bits), _p(x_mag), _p(y_mag), _st())
            return x_mag
        call("hoisdf_linear_fwd_emu", _p(x2), ldx, _p(_emu_image(W, False)), _p(b), _p(y), ldy, M, N, K, int(act),
             float(drop_p), seed, _p(bits), _st())
        return None
    call("hoisdf_linear_fwd_emu_small" if _emu_small_ok(M, x2, ldx, W, N, K) else "hoisdf_linear_fwd", _p(x2), ldx, _p(W), W.stride(0),
         _p(b), _p(y), ldy, M, N, K, int(act), float(drop_p), seed, _p(bits), _st())
    return None


def _gemm_bwd_input(dy2, lddy, bits, p, W, dx, lddx, M, N, K, accumulate, dy_mag=None):
    """-> the magnitude words of dy2 the f16x2 contraction used (given or measured here; None otherwise)"""
    if _emu_ok(M, dy2, lddy, N):
        if _h2():
            if dy_mag is None:
                dy_mag = _mag_measure(dy2, lddy, M, N)
            call("hoisdf_linear_bwd_input_emu_mag", _p(dy2), lddy, _p(bits), float(p), _p(_emu_image(W, True)), _p(dx), lddx, M, N, K,
                 int(accumulate), _p(dy_mag), None, _st())
            return dy_mag
        call("hoisdf_linear_bwd_input_emu", _p(dy2), lddy, _p(bits), float(p), _p(_emu_image(W, True)), _p(dx), lddx, M, N, K,
             int(accumulate), _st())
        return None
    call("hoisdf_linear_bwd_input_emu_small" if _emu_small_ok(M, dy2, lddy, W, N, K) else "hoisdf_linear_bwd_input", _p(dy2), lddy,
         _p(bits), float(p), _p(W), W.stride(0), _p(dx), lddx, M, N, K, int(accumulate), _st())
    return None


# grad-weight form when no magnitude words are at hand: "b3" (bf16x3, needs none) or "h2" (f16x2, the library measures both operands)
_EMU_DW_FORM = __import__("os").environ.get("HOISDF_EMU_DW_FORM", "b3")


def _gemm_bwd_weight(dy2, lddy, bits, p, x2, ldx, dW, db, M, N, K, x_scale=None, dy_scale=None, form=None, dy_mag=None, x_mag=None):
    """dW / db are zero-filled by the caller (the f32 kernel accumulates into them); the emulated form overwrites.
    form "h2" / magnitude words given (x_scale / dy_scale: what _gemm_fwd / _gemm_bwd_input returned for the same operands):
    hoisdf_linear_bwd_weight_emu_mag (f16x2 where the process runs that form)."""
    x_mag = x_mag if x_mag is not None else x_scale
    dy_mag = dy_mag if dy_mag is not None else dy_scale
    if _h2() and form != "b3":
        x_mag = x_mag if x_mag is not None else _mag_known(x2, ldx, M, K)
        dy_mag = dy_mag if dy_mag is not None else _mag_known(dy2, lddy, M, N)
    if (_GEMM_EMU and M >= _GEMM_EMU_DW_MIN_ROWS and min(N, K) >= _GEMM_EMU_DW_MIN_WIDTH and N % 4 == 0 and K % 4 == 0
            and lddy % 4 == 0 and ldx % 4 == 0 and dW.stride(0) == K and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0
            and dW.data_ptr() % 16 == 0):
        from ._lib import lib
        nws = lib().hoisdf_linear_bwd_weight_emu_workspace(M, N, K)
        ws = torch.empty(max(nws, 4), device=dW.device, dtype=torch.float32)
        if (form or _EMU_DW_FORM) == "h2" or dy_mag is not None or x_mag is not None:
            call("hoisdf_linear_bwd_weight_emu_mag", _p(dy2), lddy, _p(bits), float(p), _p(x2), ldx, _p(dW), K, _p(db), M, N, K,
                 _p(ws), nws, _p(dy_mag), _p(x_mag), _st())
        else:
            call("hoisdf_linear_bwd_weight_emu", _p(dy2), lddy, _p(bits), float(p), _p(x2), ldx, _p(dW), K, _p(db), M, N, K,
                 _p(ws), nws, _st())
        return
    if M == 0:
        return                      # an empty row set: dW / db stay zero (what the f32 entry does)
    if _GEMM_EMU and _EMU_SMALL and M <= _emu_small_max_rows():
        call("hoisdf_linear_bwd_weight_emu_small", _p(dy2), lddy, _p(bits), float(p), _p(x2), ldx, _p(dW), dW.stride(0), _p(db), M, N, K,
             _st())
        return
    ws, nws = None, 0
    if deterministic():             # partial tiles + ordered reduce instead of split-k atomics
        from ._lib import lib
        nws = lib().hoisdf_linear_bwd_weight_workspace(M, N, K)
        ws = torch.empty(max(nws, 1), device=dW.device, dtype=torch.float32) if nws > 0 else None
    call("hoisdf_linear_bwd_weight", _p(dy2), lddy, _p(bits), float(p), _p(x2), ldx, _p(dW), dW.stride(0), _p(db), M, N, K,
         _p(ws), nws, _st())


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act, drop_p, seed):
        x2 = _rows(x)
        W = W if W.stride(-1) == 1 else W.contiguous()
        _chk(x2, W, b)
        M, K = x2.shape
        N = W.shape[0]
        assert W.shape[1] == K, (W.shape, K)
        if drop_p > 0 and not act:
            raise RuntimeError("dropout without relu is not supported by linear()")
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        need_bits = bool(act) and (x.requires_grad or W.requires_grad or (b is not None and b.requires_grad))
        bits = torch.empty(M, (N + 31) // 32, device=x.device, dtype=torch.int32) if need_bits else None
        ctx.x_scale = _gemm_fwd(x2, x2.stride(0) if M > 1 else K, W, b, y, N, M, N, K, act, drop_p, seed, bits)
        ctx.save_for_backward(x2, W, bits)
        ctx.meta = (int(act), float(drop_p), b is not None, x.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, bits = ctx.saved_tensors
        act, drop_p, has_b, xshape = ctx.meta
        M, K = x2.shape
        N = W.shape[0]
        dy2 = _rows(dy)
        lddy = dy2.stride(0) if M > 1 else N
        # relu/dropout backward is fused into the staging of dy inside both contractions (1-bit sign map)
        p = drop_p if bits is not None else 0.0
        dx = dW = db = dy_scale = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            dy_scale = _gemm_bwd_input(dy2, lddy, bits, p, W, dx, K, M, N, K, 0)
            dx = dx.view(xshape)
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            buf = _zeros(N * K + (N if has_b else 0), dy.device)   # one fill (or a slice of the per-step arena)
            dW = buf[:N * K].view(N, K)
            db = buf[N * K:] if has_b else None
            _gemm_bwd_weight(dy2, lddy, bits, p, x2, x2.stride(0) if M > 1 else K, dW, db, M, N, K, ctx.x_scale, dy_scale)
        return dx, dW, db, None, None, None


def linear(x, W, b=None, act: bool = False, drop_p: float = 0.0):
    """K2/K7/K9/K11: y = dropout(relu?(x W^T + b)); fp32-exact MFMA GEMM with fused epilogue."""
    seed = next_seed() if drop_p > 0 else 0
    return _Linear.apply(x, W, b, act, drop_p, seed)


# ---------------------------------------------------------------------------------------------
# SDF decoder pieces
# ---------------------------------------------------------------------------------------------
def posenc(points: torch.Tensor) -> torch.Tensor:
    """K3 (no grad: query points are inputs). (..., 3) -> (..., 30)"""
    pts = points.reshape(-1, 3).contiguous()
    _chk(pts)
    pe = torch.empty(pts.shape[0], 30, device=pts.device, dtype=torch.float32)
    call("hoisdf_posenc_fwd", _p(pts), pts.shape[0], None, 0, 0, _p(pe), _st())
    return pe.view(*points.shape[:-1], 30)


class _WeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        v = v.contiguous()
        g = g.contiguous()
        _chk(v, g)
        out, inn = v.shape
        W = torch.empty_like(v)
        call("hoisdf_weightnorm_fwd", _p(v), _p(g), _p(W), inn, None, out, inn, _st())
        ctx.save_for_backward(v, g)
        return W

    @staticmethod
    def backward(ctx, dW):
        v, g = ctx.saved_tensors
        dW = dW.contiguous()
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        call("hoisdf_weightnorm_bwd", _p(v), _p(g), _p(dW), v.shape[1], _p(dv), _p(dg), v.shape[0], v.shape[1],
             _st())
        return dv, dg


def weight_norm(v, g):
    """W = g * v / ||v||_row  (nn.utils.weight_norm, dim=0)."""
    return _WeightNorm.apply(v, g)


class _SdfHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, b, clamp):
        h2 = _rows(h)
        w = w.reshape(-1).contiguous()
        _chk(h2, w, b)
        M, K = h2.shape
        raw = torch.empty(M, device=h.device, dtype=torch.float32)
        sdf = torch.empty(M, device=h.device, dtype=torch.float32)
        call("hoisdf_sdf_head_fwd", _p(h2), h2.stride(0) if M > 1 else K, _p(w), _p(b), _p(raw), _p(sdf), M, K,
             float(clamp), _st())
        ctx.save_for_backward(h2, w, raw)
        ctx.clamp = float(clamp)
        ctx.mark_non_differentiable(raw)
        return sdf, raw

    @staticmethod
    def backward(ctx, dsdf, _draw):
        h2, w, raw = ctx.saved_tensors
        M, K = h2.shape
        dsdf = dsdf.reshape(-1).contiguous()
        dh = torch.empty(M, K, device=dsdf.device, dtype=torch.float32)
        dwb = _zeros(K + 1, dsdf.device)
        dw, db = dwb[:K], dwb[K:]
        call("hoisdf_sdf_head_bwd", _p(dsdf), _p(raw), _p(h2), h2.stride(0) if M > 1 else K, _p(w), _p(dh), K,
             _p(dw), _p(db), M, K, ctx.clamp, _st())
        return dh, dw.view(1, K), db, None


def sdf_head(h, w, b, clamp: float):
    """tail of K4: (M,512) -> clamped sdf (M,), raw tanh (M,)"""
    return _SdfHead.apply(h, w, b, clamp)


# ---------------------------------------------------------------------------------------------
# K1-K4 behind one call: the gradient-free SDF query
# ---------------------------------------------------------------------------------------------
class SdfQueryWeights:
    """Device-side weight descriptor of ``hoisdf_sdf_query_fwd`` for one (linear_sdfin, SDFDecoder) pair: the decoder's
    weight-norm fold and the two re-laid matrices of the in-place skip-concatenation (include/hoisdf.h).  Rebuilt only
    when a parameter changed (torch's per-tensor version counters), i.e. once per optimizer step in training and once
    for a whole evaluation run."""

    def __init__(self, sdfin, decoder):
        self.sdfin, self.decoder = sdfin, decoder
        self._key = None
        self._keep = None
        self._fold = None
        self._images = None
        self.struct = None

    def _params(self):
        d = self.decoder
        ps = [l.weight for l in self.sdfin.layers] + [l.bias for l in self.sdfin.layers]
        for i in range(4):
            l = getattr(d, f"linh{i}")
            ps += [l.weight_v, l.weight_g, l.bias]
        return ps + [d.linh4.weight, d.linh4.bias]

    @torch.no_grad()
    def get(self):
        from ._lib import SdfWeights, EmuPrepItem, lib
        ps = self._params()
        key = (_WEIGHT_GEN[0], _GEMM_EMU) + tuple((p.data_ptr(), p._version) for p in ps)
        if key == self._key:
            return self.struct
        d, dev = self.decoder, ps[0].device
        s0, s1 = self.sdfin.layers[0], self.sdfin.layers[1]
        Cc = s0.weight.shape[1]
        if self._keep is None or self._keep[0].device != dev or self._keep[0].data_ptr() != s0.weight.data_ptr():
            # persistent homes of the folded / re-laid matrices: the descriptor (and the device table of the image builder) keeps
            # its addresses across optimizer steps
            z = lambda *sh: torch.zeros(*sh, device=dev)
            self._fold = dict(W0=z(512, 289), w1=z(224, 512), b1=z(224), w2=z(512, 516), W3=z(512, 512), tmp=z(512, 512))
            self._images = None
        f = self._fold
        lh = [getattr(d, f"linh{i}") for i in range(4)]
        assert lh[0].weight_v.shape == (512, 289) and lh[1].weight_v.shape == (223, 512) and lh[2].weight_v.shape == (512, 512) and \
            lh[3].weight_v.shape == (512, 512) and s0.weight.shape[0] == 512 and s1.weight.shape == (256, 512), \
            "hoisdf_sdf_query_fwd is built for the released layer sizes (512/256/223)"
        # weight-norm fold straight into the layouts of include/hoisdf.h hoisdf_sdf_weights (row 223 of w1 / b1 and the pad columns of
        # w2 stay zero: the skip concatenation is the 516-wide row itself)
        for l, out, ld in ((lh[0], f["W0"], 289), (lh[1], f["w1"], 512), (lh[2], f["tmp"], 512), (lh[3], f["W3"], 512)):
            v, g = l.weight_v.detach().contiguous(), l.weight_g.detach().contiguous()
            _chk(v, g)
            call("hoisdf_weightnorm_fwd", _p(v), _p(g), _p(out), ld, None, v.shape[0], v.shape[1], _st())
        f["w2"][:, :223] = f["tmp"][:, :223]
        f["w2"][:, 224:513] = f["tmp"][:, 223:]
        f["b1"][:223] = lh[1].bias.detach()
        keep = [s0.weight.detach().contiguous(), s0.bias.detach().contiguous(), s1.weight.detach().contiguous(),
                s1.bias.detach().contiguous(), f["W0"], lh[0].bias.detach().contiguous(), f["w1"], f["b1"], f["w2"],
                lh[2].bias.detach().contiguous(), f["W3"], lh[3].bias.detach().contiguous(),
                d.linh4.weight.detach().reshape(-1).contiguous(), d.linh4.bias.detach().contiguous()]
        _chk(*keep)
        st = SdfWeights()
        st.C = Cc
        (st.sdfin_w0, st.sdfin_b0, st.sdfin_w1, st.sdfin_b1, st.dec_w0, st.dec_b0, st.dec_w1, st.dec_b1, st.dec_w2, st.dec_b2,
         st.dec_w3, st.dec_b3, st.dec_w4, st.dec_b4) = (t.data_ptr() for t in keep)
        st.dec_ld0 = keep[4].stride(0)
        if _GEMM_EMU:
            # the bf16x3 images of the six matrices, both orientations, rebuilt by ONE launch per fold (a training step used to build
            # 48 of them one by one inside hoisdf_sdf_query_fwd / _train_fwd / _bwd)
            mats = [(keep[0], 512, Cc, Cc), (keep[2], 256, 512, 512), (keep[4], 512, 289, 289), (keep[6], 224, 512, 512),
                    (keep[8], 512, 516, 516), (keep[10], 512, 512, 512)]                # (W, N, K, ldw)
            if self._images is None:
                imgs, arr, blk = [], (EmuPrepItem * 12)(), 0
                for tr in (0, 1):
                    for i, (W, N, K, ldw) in enumerate(mats):
                        nb = lib().hoisdf_linear_emu_image_bytes(K if tr else N, N if tr else K)
                        img = torch.empty(nb, device=dev, dtype=torch.uint8)
                        it = arr[6 * tr + i]
                        it.W, it.image, it.first_block = W.data_ptr(), img.data_ptr(), blk
                        it.ldw, it.N, it.K, it.transpose = ldw, N, K, tr
                        blk += lib().hoisdf_linear_emu_prepare_blocks(N, K, tr)
                        imgs.append(img)
                table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
                self._images = (imgs, table, blk, [m_[0].data_ptr() for m_ in mats])
            imgs, table, blk, ptrs = self._images
            assert ptrs == [m_[0].data_ptr() for m_ in mats]
            call("hoisdf_linear_emu_prepare_batch", _p(table), 12, blk, _st())
            for i in range(6):
                st.emu_img[i], st.emu_img_t[i] = imgs[i].data_ptr(), imgs[6 + i].data_ptr()
        self._key, self._keep, self.struct = key, keep, st
        return st


@torch.no_grad()
def sdf_query(weights: SdfQueryWeights, pyr: "PyramidNHWC", points, center, cam_intr, scale, clamp, img_hw=(256, 256),
              sample_idx=None, feat=None, want_feat=False, want_cam=False, drop_p: float = 0.0):
    """hoisdf_sdf_query_fwd: -> (sdf (n,), sdf_raw (n,), pe (n,30), cam (n,3) | None, feat (n,C) | None).
    ``feat``: rows already gathered for these camera points (shared gather); ``want_feat``: hand the gathered rows back."""
    from ._lib import lib
    pts = points.reshape(-1, 3).contiguous()
    center, cam_intr = center.contiguous(), cam_intr.contiguous()
    _chk(pts, center, cam_intr, feat)
    n = pts.shape[0]
    dev = pts.device
    rps = points.shape[-2] if sample_idx is None else 1
    w = weights.get()
    assert w.C == pyr.C
    if feat is not None:
        assert feat.shape == (n, w.C) and feat.is_contiguous() and not want_cam
    feat_out = torch.empty(n, w.C, device=dev) if (want_feat and feat is None) else None
    nbytes = lib().hoisdf_sdf_query_workspace(n, w.C, int(feat is None and feat_out is None))
    ws = torch.empty(max(nbytes, 4) // 4, device=dev, dtype=torch.float32)
    sdf, raw = torch.empty(n, device=dev), torch.empty(n, device=dev)
    pe = torch.empty(n, 30, device=dev)
    cam = torch.empty(n, 3, device=dev) if want_cam else None
    s = pyr.struct()
    call("hoisdf_sdf_query_fwd", C.byref(s), _p(pts), _p(sample_idx), n, rps, _p(center), _p(cam_intr), float(scale),
         img_hw[0], img_hw[1], _p(feat), _p(feat_out), C.byref(w), float(clamp), float(drop_p),
         next_seed() if drop_p > 0 else 0, _p(sdf), _p(raw), _p(pe), _p(cam), _p(ws), nbytes, _st())
    return sdf, raw, pe, cam, (feat if feat is not None else feat_out)


class _SdfQueryTrain(torch.autograd.Function):
    """main/model.py:181-244 with gradients (the two SDF-loss queries of a training step) through the coarse entries
    hoisdf_sdf_query_train_fwd / hoisdf_sdf_query_bwd (csrc/sdf_query.hip): gather -> linear_sdfin -> posenc -> decoder -> head in
    one call, and the whole backward (down to the pyramid-gradient scatter) in another.  The values come from the cached folded
    weight descriptor (SdfQueryWeights); the tensor arguments after the levels only route the gradients: linear_sdfin w0, b0, w1,
    b1, then (effective weight, bias) of decoder layers 0-3 - autograd continues into the weight-norm fold -, then linh4 w, b."""

    @staticmethod
    def forward(ctx, points, center, cam_intr, scale, img_hw, clamp, drop_p, acc, wq, n_levels, *tensors):
        from ._lib import lib
        levels, params = tensors[:n_levels], tensors[n_levels:]
        assert len(params) == 14
        pyr = PyramidNHWC(levels)
        pts = points.reshape(-1, 3).contiguous()
        _chk(pts, center, cam_intr)
        n = pts.shape[0]
        rps = points.shape[-2]
        dev = pts.device
        w = wq.get()
        assert w.C == pyr.C
        seed = next_seed() if drop_p > 0 else 0
        n_saved = lib().hoisdf_sdf_query_train_saved_bytes(n, w.C)
        n_ws = lib().hoisdf_sdf_query_train_workspace_bytes(n, w.C, 0)
        saved = torch.empty(n_saved, device=dev, dtype=torch.uint8)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        sdf = torch.empty(n, device=dev)
        pe = torch.empty(n, 30, device=dev)
        cam = torch.empty(n, 3, device=dev)
        st = pyr.struct()
        call("hoisdf_sdf_query_train_fwd", C.byref(st), _p(pts), None, n, rps, _p(center), _p(cam_intr), float(scale), img_hw[0], img_hw[1],
             C.byref(w), float(clamp), float(drop_p), seed, _p(sdf), _p(pe), _p(cam), _p(saved), n_saved, _p(ws), n_ws, _st())
        ctx.save_for_backward(pts, center, cam_intr, saved)
        ctx.meta = (w, wq, float(scale), img_hw, float(clamp), float(drop_p), rps, [tuple(l.shape) for l in levels], pyr.B, acc,
                    [tuple(t.shape) for t in params])
        ctx.mark_non_differentiable(pe, cam)
        return sdf, pe, cam

    @staticmethod
    def backward(ctx, d_sdf, _dpe, _dcam):
        from ._lib import lib, SdfWeightGrads, _SDF_G
        pts, center, cam_intr, saved = ctx.saved_tensors
        w, _wq, scale, img_hw, clamp, drop_p, rps, shapes, B, acc, pshapes = ctx.meta
        dev = pts.device
        n, Cc = pts.shape[0], w.C
        sizes = [512 * Cc, 512, 256 * 512, 256, 512 * 292, 512, 224 * 512, 224, 512 * 516, 512, 512 * 512, 512, 512, 1]
        buf = _zeros(sum(sizes), dev)
        parts, off = [], 0
        for m in sizes:
            parts.append(buf[off:off + m])
            off += m
        G = SdfWeightGrads(**{k: t.data_ptr() for k, t in zip(_SDF_G, parts)})
        if acc is not None:
            grads = acc.bufs
            acc.flat.record_stream(torch.cuda.current_stream(dev))
        else:
            grads = [torch.zeros(sh, device=dev, dtype=torch.float32) for sh in shapes]
        g = Pyramid()
        g.n_levels, g.B = len(grads), B
        for i, t in enumerate(grads):
            g.data[i] = t.data_ptr()
            g.C[i], g.H[i], g.W[i] = t.shape[3], t.shape[1], t.shape[2]
        n_ws = lib().hoisdf_sdf_query_train_workspace_bytes(n, Cc, 1)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        d_sdf = d_sdf.reshape(-1).contiguous()
        call("hoisdf_sdf_query_bwd", C.byref(g), _p(pts), None, n, rps, _p(center), _p(cam_intr), scale, img_hw[0], img_hw[1], C.byref(w),
             clamp, drop_p, _p(saved), saved.numel(), _p(d_sdf), C.addressof(G), _p(ws), n_ws, _st())
        dw2p = parts[8].view(512, 516)
        pg = [parts[0].view(512, Cc), parts[1], parts[2].view(256, 512), parts[3],
              parts[4].view(512, 292)[:, :289], parts[5], parts[6].view(224, 512)[:223], parts[7][:223],
              torch.cat([dw2p[:, :223], dw2p[:, 224:513]], dim=1), parts[9], parts[10].view(512, 512), parts[11],
              parts[12].view(1, 512), parts[13]]
        pg = [t.reshape(sh) for t, sh in zip(pg, pshapes)]
        if acc is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            acc.events.append(ev)
            acc.touched = True
            lg = [None] * len(shapes)
        else:
            lg = grads
        return (None,) * 10 + tuple(lg) + tuple(pg)


_SDF_QUERY_TRAIN_C = __import__("os").environ.get("HOISDF_SDF_QUERY_TRAIN", "c") != "ops"
_TOKENS_C = __import__("os").environ.get("HOISDF_TOKENS", "c") != "ops"           # coarse K7 + K8 / K11 + K12 entries


def sdf_query_train_ok() -> bool:
    """default arithmetic only, and not while bench.py brackets the individual calls (as _coarse_layer_ok)"""
    from . import _lib
    return _SDF_QUERY_TRAIN_C and _lib._timer is None


def sdf_query_train(weights: SdfQueryWeights, pyr: "PyramidNHWC", points, center, cam_intr, scale, clamp, img_hw, drop_p, params):
    """-> (sdf clamped (n,), pe (n,30), cam (n,3)); differentiable w.r.t. the pyramid levels and the 14 routed parameters."""
    return _SdfQueryTrain.apply(points, center.contiguous(), cam_intr.contiguous(), scale, tuple(img_hw), clamp, drop_p, pyr.acc, weights,
                                len(pyr.levels), *pyr.levels, *params)


# ---------------------------------------------------------------------------------------------
# dense lattice + selection (no grad)
# ---------------------------------------------------------------------------------------------
def lattice_candidates(center, cam_intr, bbox, scale: float, bins_n: int):
    """K5: per-sample survivors of the strict bbox test, in ascending lattice order.
    Returns points (n,3), sample_idx (n,), lattice_idx (n,), counts (B,) [host list],
    offsets (B,) int32 device.  One device->host read of the B counts."""
    B = center.shape[0]
    dev = center.device
    center, cam_intr, bbox = center.contiguous(), cam_intr.contiguous(), bbox.contiguous()
    _chk(center, cam_intr, bbox)
    counts = torch.empty(B, device=dev, dtype=torch.int32)
    call("hoisdf_lattice_count", _p(center), _p(cam_intr), _p(bbox), float(scale), bins_n, B, _p(counts), _st())
    counts_h = counts.cpu()
    offsets_h = torch.zeros(B, dtype=torch.int32)
    offsets_h[1:] = torch.cumsum(counts_h, 0)[:-1].to(torch.int32)
    n = int(counts_h.sum())
    offsets = offsets_h.to(dev)
    pts = torch.empty(n, 3, device=dev, dtype=torch.float32)
    sidx = torch.empty(n, device=dev, dtype=torch.int32)
    lidx = torch.empty(n, device=dev, dtype=torch.int32)
    if n > 0:
        call("hoisdf_lattice_fill", _p(center), _p(cam_intr), _p(bbox), float(scale), bins_n, B, _p(offsets),
             _p(pts), _p(sidx), _p(lidx), _st())
    return pts, sidx, lidx, counts_h.tolist(), offsets, counts


_PINNED_I32 = {}        # B -> free page-locked int32 buffers for the survivor counts (hipHostMalloc per call would cost more than the read)


class SdfInferCounts:
    """The queued lattice-survivor count of one field (hoisdf_sdf_infer_count_begin): device counts, their page-locked host
    copy and the event behind the copy.  ``wait()`` blocks the HOST until that copy has executed - the device never drains."""

    def __init__(self, center, cam_intr, bbox, scale: float, bins_n: int):
        self.args = (center, cam_intr, bbox, float(scale), int(bins_n))      # keeps the inputs alive until the kernel has run
        B = center.shape[0]
        self.B = B
        self.counts = torch.empty(B, device=center.device, dtype=torch.int32)
        free = _PINNED_I32.setdefault(B, [])
        self.host = free.pop() if free else torch.empty(B, dtype=torch.int32, pin_memory=True)
        call("hoisdf_sdf_infer_count_begin", _p(center), _p(cam_intr), _p(bbox), float(scale), int(bins_n), B, _p(self.counts),
             C.c_void_p(self.host.data_ptr()), _st())
        self.event = torch.cuda.Event()
        self.event.record()
        self._list = None

    def matches(self, center, cam_intr, bbox, scale, bins_n) -> bool:
        a = self.args
        return (a[0].data_ptr() == center.data_ptr() and a[1].data_ptr() == cam_intr.data_ptr() and a[2].data_ptr() == bbox.data_ptr()
                and a[3] == float(scale) and a[4] == int(bins_n) and self.B == center.shape[0])

    def wait(self):
        if self._list is None:
            self.event.synchronize()
            self._list = self.host.tolist()
            _PINNED_I32[self.B].append(self.host)        # the values are copied out: the buffer can serve the next request
        return self._list


def sdf_infer_count_begin(center, cam_intr, bbox, scale: float, bins_n: int) -> SdfInferCounts:
    """queue the lattice-survivor count of sdf_infer (main/model.py:286-302; it depends on the camera inputs only) - call it
    ahead of the image encoder and hand the result to ``sdf_infer(counts=...)``: the one device -> host read of the path is
    then waited for behind work that is already queued instead of draining the pipeline."""
    center, cam_intr, bbox = center.contiguous(), cam_intr.contiguous(), bbox.contiguous()
    _chk(center, cam_intr, bbox)
    return SdfInferCounts(center, cam_intr, bbox, scale, bins_n)


@torch.no_grad()
def sdf_infer(weights: SdfQueryWeights, pyr: "PyramidNHWC", center, cam_intr, bbox, scale: float, bins_n: int, num_points: int,
              clamp: float, img_hw=(256, 256), drop_p: float = 0.0, counts: Optional[SdfInferCounts] = None):
    """hoisdf_sdf_infer_count_begin + hoisdf_sdf_infer (main/model.py:246-355 in two C-ABI calls): -> points (B,k,3), sdf (B,k),
    posenc (B,k,30) of the k = num_points lattice survivors with the smallest |sdf| per sample.  ``counts``: the survivor
    count requested earlier for the same inputs (``sdf_infer_count_begin``); without it the count is queued here and waited
    for at once.  Raises ValueError when a sample has fewer than num_points survivors (the reference fails at
    main/model.py:348)."""
    B = center.shape[0]
    dev = center.device
    center, cam_intr, bbox = center.contiguous(), cam_intr.contiguous(), bbox.contiguous()
    _chk(center, cam_intr, bbox)
    if counts is None or not counts.matches(center, cam_intr, bbox, scale, bins_n):
        counts = SdfInferCounts(center, cam_intr, bbox, scale, bins_n)
    cl = counts.wait()
    n = sum(cl)
    short = [b for b in range(B) if cl[b] < num_points]
    if short:
        raise ValueError(f"sdf_infer: sample {short[0]} has only {cl[short[0]]} lattice points inside its bbox, fewer "
                         f"than num_points={num_points} (the reference fails at main/model.py:348)")
    counts_h = (C.c_int32 * B)(*cl)
    w = weights.get()
    assert w.C == pyr.C
    nbytes = lib().hoisdf_sdf_infer_workspace(n, B, w.C)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    pts = torch.empty(B, num_points, 3, device=dev)
    sdf = torch.empty(B, num_points, device=dev)
    pe = torch.empty(B, num_points, 30, device=dev)
    s = pyr.struct()
    call("hoisdf_sdf_infer", C.byref(s), _p(center), _p(cam_intr), _p(bbox), float(scale), bins_n, B, _p(counts.counts),
         C.addressof(counts_h), num_points, img_hw[0], img_hw[1], C.byref(w), float(clamp), float(drop_p),
         next_seed() if drop_p > 0 else 0, _p(pts), _p(sdf), _p(pe), _p(ws), nbytes, _st())
    return pts, sdf, pe


def select_smallest_abs(sdf_raw, offsets, counts, k: int):
    """K6: (B,k) int32 row indices of the k smallest |sdf_raw| per sample, ascending."""
    B = offsets.shape[0]
    sel = torch.empty(B, k, device=sdf_raw.device, dtype=torch.int32)
    call("hoisdf_select_smallest_abs", _p(sdf_raw), _p(offsets), _p(counts), B, k, _p(sel), _st())
    return sel


def gather_rows(src, sel):
    src2 = _rows(src) if src.dim() > 1 else src.reshape(-1, 1)
    n = sel.numel()
    out = torch.empty(n, src2.shape[1], device=src.device, dtype=torch.float32)
    call("hoisdf_gather_rows", _p(src2), src2.stride(0), _p(sel.reshape(-1).contiguous()), n, src2.shape[1], _p(out),
         src2.shape[1], _st())
    return out


# ---------------------------------------------------------------------------------------------
# tokens
# ---------------------------------------------------------------------------------------------
class _TokenBuild(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, cam, center, pe, feat, sdf, beta, row0):
        """writes rows [row0, row0+P) of every sample of the batch-first token buffer tok (B,S,D)"""
        B, S, D = tok.shape
        P = cam.shape[0] // B
        feat2 = _rows(feat)
        cam, pe, sdf = cam.contiguous(), pe.reshape(-1, 30).contiguous(), sdf.reshape(-1).contiguous()
        _chk(tok, cam, center, pe, feat2, sdf, beta)
        call("hoisdf_token_build_fwd", _p(cam), _p(center), _p(pe), _p(feat2), feat2.stride(0), _p(sdf), _p(beta),
             _p(tok), B, P, S, row0, D, _st())
        ctx.save_for_backward(feat2, sdf, beta)
        ctx.meta = (B, P, S, row0, D, feat.shape)
        ctx.mark_dirty(tok)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        feat2, sdf, beta = ctx.saved_tensors
        B, P, S, row0, D, fshape = ctx.meta
        dtok = dtok.contiguous()
        dfeat = torch.empty(B * P, D - 33, device=dtok.device, dtype=torch.float32)
        dbeta = _zeros(1, dtok.device)
        part = torch.empty(lib().hoisdf_token_build_bwd_partials(), device=dtok.device, dtype=torch.float32)
        call("hoisdf_token_build_bwd_ordered", _p(dtok), _p(feat2), feat2.stride(0), _p(sdf), _p(beta), _p(dfeat), D - 33,
             _p(dbeta), _p(part), B, P, S, row0, D, _st())       # beta gradient summed in block order: no float atomics
        # the rows this op wrote do not depend on the incoming buffer contents
        dtok_in = dtok.clone()
        dtok_in[:, row0:row0 + P] = 0
        return dtok_in, None, None, None, dfeat.view(fshape), None, dbeta, None


def token_build(tok, cam, center, pe, feat, sdf, beta, row0: int):
    """K8: tok[b, row0+p] = [cam-center | pe | feat * sigmoid(sdf/beta)/beta]."""
    return _TokenBuild.apply(tok, cam, center.contiguous(), pe, feat, sdf, beta, row0)


# ---------------------------------------------------------------------------------------------
# attention + layer norm
# ---------------------------------------------------------------------------------------------
def _attn_fwd(q, k, v, H, kv_len, drop_p, seed):
    B, Lq, E = q.shape
    Lk = k.shape[1]
    for t, L in ((q, Lq), (k, Lk), (v, Lk)):
        assert t.stride(2) == 1 and t.stride(0) == L * t.stride(1), "attention operands must be row-uniform views"
    o = torch.empty(B, Lq, E, device=q.device, dtype=torch.float32)
    lse = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    call("hoisdf_attention_fwd", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(lse),
         B, H, Lq, Lk, kv_len, float(drop_p), seed, _st())
    return o, lse


def _attn_bwd(q, k, v, o, lse, do, dq, dk, dv, H, kv_len, drop_p, seed):
    B, Lq, E = q.shape
    Lk = k.shape[1]
    assert dq.stride(1) == q.stride(1) and dk.stride(1) == k.stride(1) and dv.stride(1) == v.stride(1)
    delta = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    call("hoisdf_attention_bwd", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(do), E,
         _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, H, Lq, Lk, kv_len, drop_p, seed, _st())


_ATTENTION_EMU = __import__("os").environ.get("HOISDF_ATTENTION", "emu") != "f32"


def set_attention_emu(on: bool) -> None:
    """cfg.attention_emu (default on; HOISDF_ATTENTION=f32 turns it off): the large attention calls (forward with dropout + LSE,
    fused one-pass backward) as fp32 emulated on the bf16 MFMA pipe - exact three-way bf16 splits of Q, K, V, dO, P and dS, six
    products per product, f32 accumulation / softmax (csrc/attention_emu.hip).  fp32-equivalent results, no atomics.  Off: the
    exact-f32 MFMA kernels of csrc/attention.hip."""
    global _ATTENTION_EMU
    _ATTENTION_EMU = bool(on)


def attention_emu() -> bool:
    return _ATTENTION_EMU


def _use_split(Lq: int) -> int:
    """attention kernel family for a call with Lq queries: 0 = exact-f32 MFMA, 2 = bf16x3 emulated fp32 (default; family 1, the
    f16 hi + lo split-precision training kernels of round 2, was retired in round 4: the emulated fp32 kernels are faster and
    exact).  The 17-query decoder attention keeps its own f32 kernels."""
    if Lq < 32:
        return 0
    return 2 if _ATTENTION_EMU else 0


# forward workspaces whose Q / K / V planes the matching backward reuses (hoisdf_attention_fwd_emu(keep = 1) ->
# hoisdf_attention_bwd_emu(fwd_workspace)): keyed by the operands' addresses - the autograd node keeps q, k, v alive until its
# backward, so a key cannot be taken over by another live call; leftovers of graphs that never ran backward go at the next step /
# at 64 entries.  HOISDF_PLANES_KEEP=0 (or the older name HOISDF_SPLIT_KEEP=0): every backward converts on its own.
_SPLIT_KEEP = __import__("os").environ.get("HOISDF_PLANES_KEEP", __import__("os").environ.get("HOISDF_SPLIT_KEEP", "1")) != "0"


def _planes_key(q, k, v, H, kv_len):
    return (q.data_ptr(), k.data_ptr(), v.data_ptr(), tuple(q.shape), k.shape[1], H, kv_len)


_EMU_PLANES = {}


_ATTN_FORM_H2 = __import__("os").environ.get("HOISDF_ATTN_FORM", "h2")[:1].lower() != "b"


def _attn_h2(rows: int) -> bool:
    """the encoder layers' attention forward runs in the f16x2 form (csrc/layers.hip geometry(): where their linear layers do)"""
    return _ATTN_FORM_H2 and _GEMM_EMU and _h2() and rows >= _GEMM_EMU_MIN_ROWS


def _attn_fwd_emu(q, k, v, H, kv_len, drop_p, seed, keep=False, heads=None):
    """keep: a backward will follow - convert Q, K, V once into every plane it needs and park the workspace for it.
    heads = (q, k, v head magnitudes: one scale per (sample, head) and operand) -> the f16x2 form of the forward (no planes kept)"""
    from ._lib import lib
    B, Lq, E = q.shape
    Lk = k.shape[1]
    for t, L in ((q, Lq), (k, Lk), (v, Lk)):
        assert t.stride(2) == 1 and t.stride(0) == L * t.stride(1), "attention operands must be row-uniform views"
    keep = keep and _SPLIT_KEEP
    nbytes = lib().hoisdf_attention_emu_workspace(B, H, Lq, Lk, 2 if keep else 0)
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    o = torch.empty(B, Lq, E, device=q.device, dtype=torch.float32)
    lse = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    if heads is not None:
        call("hoisdf_attention_fwd_emu_mag", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(lse), B, H, Lq, Lk,
             kv_len, float(drop_p), seed, _p(ws), nbytes, 0, _p(heads[0]), _p(heads[1]), _p(heads[2]), None, _st())
        return o, lse
    call("hoisdf_attention_fwd_emu", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(lse), B, H, Lq, Lk,
         kv_len, float(drop_p), seed, _p(ws), nbytes, int(keep), _st())
    if keep:
        if len(_EMU_PLANES) >= 64:
            _EMU_PLANES.clear()
        _EMU_PLANES[_planes_key(q, k, v, H, kv_len)] = ws
    return o, lse


def _attn_bwd_emu(q, k, v, o, lse, do, dq, dk, dv, H, kv_len, drop_p, seed, heads=None):
    """heads: the (q, k, v) head magnitudes the forward used -> the f16x2 form of the backward"""
    from ._lib import lib
    B, Lq, E = q.shape
    Lk = k.shape[1]
    assert dq.stride(1) == q.stride(1) and dk.stride(1) == k.stride(1) and dv.stride(1) == v.stride(1)
    kept = _EMU_PLANES.pop(_planes_key(q, k, v, H, kv_len), None)
    nbytes = lib().hoisdf_attention_bwd_emu_workspace(B, H, Lq, Lk, int(kept is not None))
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    delta = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    if kept is not None:
        kept.record_stream(torch.cuda.current_stream(q.device))
    if heads is not None:
        do_hm = _head_measure(do, E, B * Lq, H, Lq)
        call("hoisdf_attention_bwd_emu_mag", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(do), E, _p(lse),
             _p(delta), _p(dq), _p(dk), _p(dv), B, H, Lq, Lk, kv_len, float(drop_p), seed, _p(kept), _p(ws), nbytes, _p(heads[0]),
             _p(heads[1]), _p(heads[2]), _p(do_hm), None, _st())
        return
    call("hoisdf_attention_bwd_emu", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, _p(do), E, _p(lse),
         _p(delta), _p(dq), _p(dk), _p(dv), B, H, Lq, Lk, kv_len, float(drop_p), seed, _p(kept), _p(ws), nbytes, _st())


_ATTN_BWD_EMU = __import__("os").environ.get("HOISDF_ATTN_BWD", "emu") != "f32"


def _emu_bwd() -> bool:
    """Emulated attention calls run their backward emulated as well (default; HOISDF_ATTN_BWD=f32 keeps the exact-f32 fused
    backward next to the emulated forward): the 8-wave form of csrc/attention_emu.hip (16 keys per wave, two waves per SIMD)
    measures 2.2 ms against 3.07 ms for the f32 kernel at B = 32, S = 2048 (tools/mb_attn_emu.py; the first, one-wave-per-SIMD
    form was 3.4 ms).  It is order-fixed (no atomics), so it is also what deterministic mode uses."""
    return _ATTN_BWD_EMU or deterministic()


def _attn_fwd_mode(mode, q, k, v, H, kv_len, drop_p, seed, keep=False, heads=None):
    if mode == 2 and heads is not None:
        return _attn_fwd_emu(q, k, v, H, kv_len, drop_p, seed, heads=heads)
    if mode == 2:
        return _attn_fwd_emu(q, k, v, H, kv_len, drop_p, seed, keep=keep and _emu_bwd())
    return _attn_fwd(q, k, v, H, kv_len, drop_p, seed)


def _attn_bwd_mode(mode, *a, heads=None):
    if mode == 2 and _emu_bwd():
        return _attn_bwd_emu(*a, heads=heads)
    return _attn_bwd(*a)


class _AttentionSelf(torch.autograd.Function):
    """qkv (B,L,3E): the packed in-projection output [q | k | v]."""

    @staticmethod
    def forward(ctx, qkv, H, kv_len, drop_p, seed):
        qkv = qkv.contiguous()
        _chk(qkv)
        E = qkv.shape[2] // 3
        split = _use_split(qkv.shape[1])
        o, lse = _attn_fwd_mode(split, qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, kv_len, drop_p, seed,
                                keep=any(ctx.needs_input_grad))
        ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (H, kv_len, float(drop_p), seed, split)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        H, kv_len, drop_p, seed, split = ctx.meta
        E = qkv.shape[2] // 3
        d = torch.empty_like(qkv)
        _attn_bwd_mode(split, qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], o, lse, do.contiguous(), d[:, :, :E],
                       d[:, :, E:2 * E], d[:, :, 2 * E:], H, kv_len, drop_p, seed)
        return d, None, None, None, None


class _AttentionCross(torch.autograd.Function):
    """q (B,Lq,E) contiguous; kv (B,Lk,2E): the packed [k | v] projection of the memory."""

    @staticmethod
    def forward(ctx, q, kv, H, kv_len, drop_p, seed):
        q, kv = q.contiguous(), kv.contiguous()
        _chk(q, kv)
        E = q.shape[2]
        split = _use_split(q.shape[1])
        o, lse = _attn_fwd_mode(split, q, kv[:, :, :E], kv[:, :, E:], H, kv_len, drop_p, seed, keep=any(ctx.needs_input_grad))
        ctx.save_for_backward(q, kv, o, lse)
        ctx.meta = (H, kv_len, float(drop_p), seed, split)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        H, kv_len, drop_p, seed, split = ctx.meta
        E = q.shape[2]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        _attn_bwd_mode(split, q, kv[:, :, :E], kv[:, :, E:], o, lse, do.contiguous(), dq, dkv[:, :, :E], dkv[:, :, E:], H, kv_len,
                       drop_p, seed)
        return dq, dkv, None, None, None, None


_ATTENTION_F16_EVAL = False


def set_attention_f16_eval(on: bool) -> None:
    """BASELINE configs[4] ("fp16 MFMA attention"): route gradient-free, dropout-free attention calls through the f16
    MFMA kernel (f32 accumulation / softmax).  Off by default: the f32 kernel is the parity configuration."""
    global _ATTENTION_F16_EVAL
    _ATTENTION_F16_EVAL = bool(on)


def _attn_fwd_f16(q, k, v, H, kv_len):
    B, Lq, E = q.shape
    Lk = k.shape[1]
    for t, L in ((q, Lq), (k, Lk), (v, Lk)):
        assert t.stride(2) == 1 and t.stride(0) == L * t.stride(1), "attention operands must be row-uniform views"
    from ._lib import lib
    o = torch.empty(B, Lq, E, device=q.device, dtype=torch.float32)
    if _ATTN16_LEGACY:              # HOISDF_ATTN16=f16: round 2's f16 hi + lo kernel (A/B runs)
        nbytes = lib().hoisdf_attention_f16_workspace(B, H, Lk)
        ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
        call("hoisdf_attention_fwd_f16", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, B, H, Lq, Lk,
             kv_len, _p(ws), nbytes, _st())
        return o
    # round 5: bf16 hi + lo operands on the pipelined forward (hoisdf_attention_fwd_bf16x2)
    nbytes = lib().hoisdf_attention_bf16x2_workspace(B, H, Lq, Lk)
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    call("hoisdf_attention_fwd_bf16x2", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), E, B, H, Lq, Lk,
         kv_len, _p(ws), nbytes, _st())
    return o


_ATTN16_LEGACY = __import__("os").environ.get("HOISDF_ATTN16", "bf16x2") == "f16"


def _use_f16(drop_p, *tensors) -> bool:
    return _ATTENTION_F16_EVAL and drop_p == 0.0 and not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))


def attention_self(qkv, H: int, kv_len: Optional[int] = None, drop_p: float = 0.0):
    """K9: streaming-softmax self-attention on a packed (B,L,3E) projection; q scaled by 1/8 inside."""
    kv_len = qkv.shape[1] if kv_len is None else kv_len
    if _use_f16(drop_p, qkv):
        qkv = qkv.contiguous()
        E = qkv.shape[2] // 3
        return _attn_fwd_f16(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, kv_len)
    seed = next_seed() if drop_p > 0 else 0
    return _AttentionSelf.apply(qkv, H, kv_len, drop_p, seed)


def attention_cross(q, kv, H: int, kv_len: Optional[int] = None, drop_p: float = 0.0):
    """K10: cross-attention of q (B,Lq,E) over a packed (B,Lk,2E) memory projection; only keys
    < kv_len are attended (memory_mask of common/utils/misc.py:34-47)."""
    kv_len = kv.shape[1] if kv_len is None else kv_len
    if _use_f16(drop_p, q, kv):
        q, kv = q.contiguous(), kv.contiguous()
        E = q.shape[2]
        return _attn_fwd_f16(q, kv[:, :, :E], kv[:, :, E:], H, kv_len)
    seed = next_seed() if drop_p > 0 else 0
    return _AttentionCross.apply(q, kv, H, kv_len, drop_p, seed)


class _AttentionSmall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8, H, drop_p, seed):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        _chk(q, k, v)
        B, Lq, E = q.shape
        Lk = k.shape[1]
        o = torch.empty(B, Lq, E, device=q.device, dtype=torch.float32)
        probs = torch.empty(B, H, Lq, Lk, device=q.device, dtype=torch.float32)
        call("hoisdf_attention_small_fwd", _p(q), E, _p(k), E, _p(v), E, _p(mask_u8), _p(o), E, _p(probs), B, H, Lq,
             Lk, float(drop_p), seed, _st())
        ctx.save_for_backward(q, k, v, probs)
        ctx.meta = (H, float(drop_p), seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, probs = ctx.saved_tensors
        H, drop_p, seed = ctx.meta
        B, Lq, E = q.shape
        Lk = k.shape[1]
        do = do.contiguous()
        dq = torch.empty_like(q)
        dk = torch.zeros_like(k)
        dv = torch.zeros_like(v)
        call("hoisdf_attention_small_bwd", _p(q), E, _p(k), E, _p(v), E, _p(probs), _p(do), E, _p(dq), _p(dk), _p(dv),
             B, H, Lq, Lk, drop_p, seed, _st())
        return dq, dk, dv, None, None, None, None


def attention_small(q, k, v, mask_u8, H: int, drop_p: float = 0.0):
    """masked attention for the 17 MANO queries (Lq, Lk <= 64); mask (Lq,Lk) uint8, 1 = masked."""
    seed = next_seed() if drop_p > 0 else 0
    return _AttentionSmall.apply(q, k, v, mask_u8, H, drop_p, seed)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, drop_p, seed):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        r2 = None if r is None else r.reshape(-1, x.shape[-1]).contiguous()
        _chk(x2, r2, gamma, beta)
        M, D = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        call("hoisdf_add_layernorm_fwd", _p(x2), _p(r2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D,
             float(eps), float(drop_p), seed, _st())
        ctx.save_for_backward(x2, r2, gamma, mean, rstd)
        ctx.meta = (float(drop_p), seed, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, gamma, mean, rstd = ctx.saved_tensors
        drop_p, seed, shape = ctx.meta
        M, D = x2.shape
        dy2 = dy.reshape(M, D).contiguous()
        dx = torch.empty_like(x2)
        dr = None if r2 is None else torch.empty_like(x2)
        dgb = _zeros(2 * D, dy.device).view(2, D)
        dg, db = dgb[0], dgb[1]
        call("hoisdf_add_layernorm_bwd", _p(dy2), _p(x2), _p(r2), _p(gamma), _p(mean), _p(rstd), None, _p(dx), _p(dr),
             _p(dg), _p(db), M, D, drop_p, seed, _st())
        return dx.view(shape), (None if dr is None else dr.view(shape)), dg, db, None, None, None


def add_layernorm(x, r, gamma, beta, eps: float = 1e-5, drop_p: float = 0.0):
    """LN(x + dropout(r)); r=None -> plain LayerNorm."""
    seed = next_seed() if (drop_p > 0 and r is not None) else 0
    return _AddLayerNorm.apply(x, r, gamma, beta, eps, drop_p if r is not None else 0.0, seed)


class _ResidualDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, drop_p, seed):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        r2 = r.reshape(-1, x.shape[-1]).contiguous()
        _chk(x2, r2)
        M, D = x2.shape
        y = torch.empty_like(x2)
        call("hoisdf_residual_dropout", _p(x2), _p(r2), _p(y), M, D, float(drop_p), seed, _st())
        ctx.meta = (float(drop_p), seed, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        drop_p, seed, shape = ctx.meta
        if drop_p == 0.0:
            return dy, dy, None, None
        dy2 = dy.reshape(-1, shape[-1]).contiguous()
        dr = torch.empty_like(dy2)
        call("hoisdf_residual_dropout", None, _p(dy2), _p(dr), dy2.shape[0], dy2.shape[1], drop_p, seed, _st())
        return dy, dr.view(shape), None, None


def residual_dropout(x, r, drop_p: float = 0.0):
    """x + dropout(r): the residual of a pre-norm layer (cfg.pre_norm; the post-norm layers use add_layernorm)."""
    return _ResidualDropout.apply(x, r, drop_p, next_seed() if drop_p > 0 else 0)


# ---------------------------------------------------------------------------------------------
# one transformer encoder layer as ONE autograd node
# ---------------------------------------------------------------------------------------------
def _lin_fwd(x2, W, b, act, drop_p, seed, need_bits, out=None):
    M, K = x2.shape
    N = W.shape[0]
    y = torch.empty(M, N, device=x2.device, dtype=torch.float32) if out is None else out
    bits = torch.empty(M, (N + 31) // 32, device=x2.device, dtype=torch.int32) if (act and need_bits) else None
    _gemm_fwd(x2, x2.stride(0), W, b, y, y.stride(0), M, N, K, act, drop_p, seed, bits)
    return y, bits


def _lin_bwd_input(dy2, bits, p, W, dx, accumulate):
    M, N = dy2.shape
    K = W.shape[1]
    _gemm_bwd_input(dy2, dy2.stride(0), bits, p if bits is not None else 0.0, W, dx, dx.stride(0), M, N, K, accumulate)


def _lin_bwd_weight(dy2, bits, p, x2, dW, db):
    M, N = dy2.shape
    K = x2.shape[1]
    _gemm_bwd_weight(dy2, dy2.stride(0), bits, p if bits is not None else 0.0, x2, x2.stride(0), dW, db, M, N, K)


class _EncoderLayer(torch.autograd.Function):
    """common/nets/transformer.py:286-302 (TransformerEncoderLayer.forward_post) + the stack's ``inter_norm`` of the layer
    output (:117-131) as ONE autograd node: the same kernels as the op-by-op path, but the gradients of tensors with several
    consumers (x: attention projection + residual; x1: FFN + residual; x2: next layer + inter_norm) are accumulated by
    the kernels themselves (``accumulate`` epilogue of the grad-input GEMM, ``dx_add`` of the LayerNorm backward) instead
    of by separate autograd add kernels, and every weight-gradient buffer of the layer comes from one arena slice.
    x (B,S,E); n_query < S: only the first n_query rows are produced (keys / values still come from all S rows)."""

    @staticmethod
    def forward(ctx, x, n_query, p, H, w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3, be3, eps, n_inter=None):
        x = x.contiguous()
        _chk(x, w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3, be3)
        B, S, E = x.shape
        nq = S if (n_query is None or n_query >= S) else int(n_query)
        full = nq == S
        dev = x.device
        x2d = x.view(B * S, E)
        s_attn = next_seed() if p > 0 else 0
        s_ln1 = next_seed() if p > 0 else 0
        s_ffn = next_seed() if p > 0 else 0
        s_ln2 = next_seed() if p > 0 else 0
        split = _use_split(nq)
        if full:
            qkv, _ = _lin_fwd(x2d, w_in, b_in, False, 0.0, 0, False)
            qkv3 = qkv.view(B, S, 3 * E)
            q, k, v = qkv3[:, :, :E], qkv3[:, :, E:2 * E], qkv3[:, :, 2 * E:]
            xq = x
        else:
            xq = x[:, :nq].contiguous()
            qbuf, _ = _lin_fwd(xq.view(B * nq, E), w_in[:E], b_in[:E], False, 0.0, 0, False)
            kvbuf, _ = _lin_fwd(x2d, w_in[E:], b_in[E:], False, 0.0, 0, False)
            q = qbuf.view(B, nq, E)
            kv3 = kvbuf.view(B, S, 2 * E)
            k, v = kv3[:, :, :E], kv3[:, :, E:]
        heads = None
        if _use_f16(p, x, w_in, w_out, w1, w2):
            # gradient-free eval with cfg.attention_f16_eval: the f16-operand kernel (no LSE: nothing is saved for a backward)
            o, lse = _attn_fwd_f16(q, k, v, H, S), None
        else:
            if split == 2 and _attn_h2(B * nq) and E == 64 * H:
                # the f16x2 form of the forward, as the coarse entry runs it: one scale per (sample, head) from the projected matrices
                if full:
                    hm = _head_measure(qkv, 3 * E, B * S, 3 * H, S)
                    heads = (hm, hm[H * B:], hm[2 * H * B:])
                else:
                    hq, hkv = _head_measure(qbuf, E, B * nq, H, nq), _head_measure(kvbuf, 2 * E, B * S, 2 * H, S)
                    heads = (hq, hkv, hkv[H * B:])
            o, lse = _attn_fwd_mode(split, q, k, v, H, S, p, s_attn, keep=any(ctx.needs_input_grad), heads=heads)
        M = B * nq
        a, _ = _lin_fwd(o.view(M, E), w_out, b_out, False, 0.0, 0, False)
        xq2 = xq.view(M, E)
        x1 = torch.empty(M, E, device=dev)
        st = torch.empty(6, M, device=dev)                     # mean / rstd of the three LayerNorms
        call("hoisdf_add_layernorm_fwd", _p(xq2), _p(a), _p(g1), _p(be1), _p(x1), _p(st[0]), _p(st[1]), M, E, float(eps),
             float(p), s_ln1, _st())
        h, bits = _lin_fwd(x1, w1, b1, True, p, s_ffn, True)
        f, _ = _lin_fwd(h, w2, b2, False, 0.0, 0, False)
        x2 = torch.empty(M, E, device=dev)
        call("hoisdf_add_layernorm_fwd", _p(x1), _p(f), _p(g2), _p(be2), _p(x2), _p(st[2]), _p(st[3]), M, E, float(eps),
             float(p), s_ln2, _st())
        # inter_norm of the layer output: only rows < n_inter of every sample are ever read by the caller
        ni = nq if (n_inter is None or n_inter >= nq) else int(n_inter)
        y = torch.empty(B * ni, E, device=dev)
        if ni == nq:
            call("hoisdf_add_layernorm_fwd", _p(x2), None, _p(g3), _p(be3), _p(y), _p(st[4]), _p(st[5]), M, E, float(eps),
                 0.0, 0, _st())
        else:
            call("hoisdf_layernorm_rows_fwd", _p(x2), _p(g3), _p(be3), _p(y), _p(st[4]), _p(st[5]), B, nq, ni, E, float(eps),
                 _st())
        ctx.save_for_backward(x, qkv if full else qbuf, qkv if full else kvbuf, o, lse, a, x1, h, bits, f, x2, st, w_in,
                              w_out, w1, w2, g1, g2, g3)
        ctx.meta = (B, S, E, nq, full, float(p), H, (s_attn, s_ln1, s_ffn, s_ln2), split, ni)
        ctx.attn_heads = heads
        return x2.view(B, nq, E), y.view(B, ni, E)

    @staticmethod
    def backward(ctx, g_x2, g_y):
        (x, qs, ks, o, lse, a, x1, h, bits, f, x2, st, w_in, w_out, w1, w2, g1, g2, g3) = ctx.saved_tensors
        B, S, E, nq, full, p, H, (s_attn, s_ln1, s_ffn, s_ln2), split, ni = ctx.meta
        dev = x.device
        M, F = B * nq, w1.shape[0]
        # one zero-initialised slice for every parameter gradient of the layer
        sizes = [3 * E * E, 3 * E, E * E, E, E, E, F * E, F, E * F, E, E, E, E, E]
        buf = _zeros(sum(sizes), dev)
        parts, off = [], 0
        for n in sizes:
            parts.append(buf[off:off + n])
            off += n
        dw_in, db_in, dw_out, db_out, dg1, dbe1, dw1, db1, dw2, db2, dg2, dbe2, dg3, dbe3 = parts
        dw_in, dw_out, dw1, dw2 = dw_in.view(3 * E, E), dw_out.view(E, E), dw1.view(F, E), dw2.view(E, F)
        gx2 = None if g_x2 is None else g_x2.contiguous().view(M, E)
        dx2 = torch.empty(M, E, device=dev)
        if g_y is not None:
            gy = g_y.contiguous().view(B * ni, E)
            if ni == nq:
                call("hoisdf_add_layernorm_bwd", _p(gy), _p(x2), None, _p(g3), _p(st[4]), _p(st[5]), _p(gx2), _p(dx2), None,
                     _p(dg3), _p(dbe3), M, E, 0.0, 0, _st())
            else:
                call("hoisdf_layernorm_rows_bwd", _p(gy), _p(x2), _p(g3), _p(st[4]), _p(st[5]), _p(gx2), _p(dx2), _p(dg3),
                     _p(dbe3), B, nq, ni, E, _st())
        else:
            dx2 = gx2
        dx1 = torch.empty(M, E, device=dev)
        df = torch.empty(M, E, device=dev)
        call("hoisdf_add_layernorm_bwd", _p(dx2), _p(x1), _p(f), _p(g2), _p(st[2]), _p(st[3]), None, _p(dx1), _p(df),
             _p(dg2), _p(dbe2), M, E, p, s_ln2, _st())
        dh = torch.empty(M, F, device=dev)
        _lin_bwd_input(df, None, 0.0, w2, dh, False)
        _lin_bwd_weight(df, None, 0.0, h, dw2, db2)
        _lin_bwd_input(dh, bits, p, w1, dx1, True)                   # dx1 += : the FFN branch joins the residual branch
        _lin_bwd_weight(dh, bits, p, x1, dw1, db1)
        xq2 = (x if full else x[:, :nq].contiguous()).view(M, E)
        dxq = torch.empty(M, E, device=dev)
        da = torch.empty(M, E, device=dev)
        call("hoisdf_add_layernorm_bwd", _p(dx1), _p(xq2), _p(a), _p(g1), _p(st[0]), _p(st[1]), None, _p(dxq), _p(da),
             _p(dg1), _p(dbe1), M, E, p, s_ln1, _st())
        do = torch.empty(M, E, device=dev)
        _lin_bwd_input(da, None, 0.0, w_out, do, False)
        _lin_bwd_weight(da, None, 0.0, o.view(M, E), dw_out, db_out)
        heads = ctx.attn_heads                                        # (the f16x2 forward's head magnitudes: the backward runs in the same form)
        bwd = lambda *a_: _attn_bwd_mode(split, *a_, heads=heads)
        do3 = do.view(B, nq, E)
        if full:
            qkv3 = qs.view(B, S, 3 * E)
            dqkv = torch.empty(B, S, 3 * E, device=dev)
            bwd(qkv3[:, :, :E], qkv3[:, :, E:2 * E], qkv3[:, :, 2 * E:], o, lse, do3,
                dqkv[:, :, :E], dqkv[:, :, E:2 * E], dqkv[:, :, 2 * E:], H, S, p, s_attn)
            d2 = dqkv.view(B * S, 3 * E)
            _lin_bwd_input(d2, None, 0.0, w_in, dxq, True)           # dx += : attention branch joins the residual branch
            _lin_bwd_weight(d2, None, 0.0, x.view(B * S, E), dw_in, db_in)
            dx = dxq.view(B, S, E)
        else:
            kv3 = ks.view(B, S, 2 * E)
            dq = torch.empty(B, nq, E, device=dev)
            dkv = torch.empty(B, S, 2 * E, device=dev)
            bwd(qs.view(B, nq, E), kv3[:, :, :E], kv3[:, :, E:], o, lse, do3, dq, dkv[:, :, :E], dkv[:, :, E:], H, S, p,
                s_attn)
            _lin_bwd_input(dq.view(M, E), None, 0.0, w_in[:E], dxq, True)
            _lin_bwd_weight(dq.view(M, E), None, 0.0, xq2, dw_in[:E], db_in[:E])
            dxf = torch.empty(B * S, E, device=dev)
            _lin_bwd_input(dkv.view(B * S, 2 * E), None, 0.0, w_in[E:], dxf, False)
            _lin_bwd_weight(dkv.view(B * S, 2 * E), None, 0.0, x.view(B * S, E), dw_in[E:], db_in[E:])
            dx = dxf.view(B, S, E)
            dx[:, :nq] += dxq.view(B, nq, E)
        return (dx, None, None, None, dw_in, db_in, dw_out, db_out, dg1, dbe1, dw1, db1, dw2, db2, dg2, dbe2, dg3, dbe3, None, None)


class _EncoderLayerC(torch.autograd.Function):
    """The same layer through the coarse C entries hoisdf_encoder_layer_fwd / _bwd (csrc/layers.hip): the chain of
    ``_EncoderLayer`` issued by the library itself - one C-ABI call, one saved-activation buffer and one scratch buffer per
    direction instead of ~15 calls and ~20 allocations.  Same kernels in the same order, same dropout seeds."""

    @staticmethod
    def _images(w, names, transpose, rows_q, rows_s, E, full):
        """cached bf16x3 images of the weights the library would otherwise rebuild in its workspace on every call"""
        if not _GEMM_EMU:
            return
        w_in, w_out, w1, w2 = names
        sfx = "img_t_" if transpose else "img_"
        big_q, big_s = rows_q >= _GEMM_EMU_MIN_ROWS, rows_s >= _GEMM_EMU_MIN_ROWS
        if full and big_s:
            setattr(w, sfx + "in", _emu_image(w_in, transpose).data_ptr())
        if not full:
            if big_q:
                setattr(w, sfx + "in_q", _emu_image(w_in[:E], transpose).data_ptr())
            if big_s:
                setattr(w, sfx + "in_kv", _emu_image(w_in[E:], transpose).data_ptr())
        if big_q:
            setattr(w, sfx + "out", _emu_image(w_out, transpose).data_ptr())
            setattr(w, sfx + "1", _emu_image(w1, transpose).data_ptr())
            setattr(w, sfx + "2", _emu_image(w2, transpose).data_ptr())

    @staticmethod
    def forward(ctx, x, n_query, p, H, w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3, be3, eps, n_inter=None):
        from ._lib import lib, EncoderLayerDesc, EncoderLayerWeights
        x = x.contiguous()
        params = (w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3, be3)
        _chk(x, *params)
        assert all(t.is_contiguous() for t in params)
        B, S, E = x.shape
        nq = S if (n_query is None or n_query >= S) else int(n_query)
        ni = nq if (n_inter is None or n_inter >= nq) else int(n_inter)
        d = EncoderLayerDesc(B=B, S=S, E=E, F=w1.shape[0], H=H, n_query=nq, n_inter=ni, eps=eps, drop_p=p,
                             attention=2 if _use_split(nq) == 2 else 0, attention_bwd_emulated=int(_emu_bwd() and _SPLIT_KEEP),
                             training=int(any(ctx.needs_input_grad)))
        for i in range(4):                                     # attention, after out-projection, FFN hidden, after the FFN
            d.seed[i] = next_seed() if p > 0 else 0
        # magnitude words of x (f16x2 form): left by the layer below next to its saved activations (kept alive with them)
        xm = getattr(x, "_hoisdf_mag", None)
        if xm is not None and xm[2] == x._version:
            d.x_mag = xm[0]
            ctx.x_mag_owner = xm[1]
        w = EncoderLayerWeights(**{n: t.data_ptr() for n, t in zip(_ENC_W_NAMES, params)})
        _EncoderLayerC._images(w, (w_in, w_out, w1, w2), False, B * nq, B * S, E, nq == S)
        dp, wp = C.addressof(d), C.addressof(w)
        n_saved = lib().hoisdf_encoder_layer_saved_bytes(dp) if d.training else 0
        n_ws = lib().hoisdf_encoder_layer_workspace_bytes(dp, 0)
        dev = x.device
        saved = torch.empty(n_saved, device=dev, dtype=torch.uint8) if n_saved else None
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        x2 = torch.empty(B, nq, E, device=dev)
        y = torch.empty(B, ni, E, device=dev)
        call("hoisdf_encoder_layer_fwd", _p(x), wp, dp, _p(x2), _p(y), _p(saved), n_saved, _p(ws), n_ws, _st())
        if saved is not None:
            om = lib().hoisdf_encoder_layer_out_mag(dp, _p(saved))
            if om:
                x2._hoisdf_mag = (om, saved, x2._version)
        ctx.save_for_backward(x, x2, saved, *params)
        ctx.desc = d
        return x2, y

    @staticmethod
    def backward(ctx, g_x2, g_y):
        from ._lib import lib, EncoderLayerWeights, EncoderLayerGrads
        x, x2, saved, *params = ctx.saved_tensors
        w_in, _, w_out, _, _, _, w1, _, w2 = params[:9]
        d = ctx.desc
        B, S, E, F, nq = d.B, d.S, d.E, d.F, d.n_query
        dev = x.device
        sizes = [3 * E * E, 3 * E, E * E, E, E, E, F * E, F, E * F, E, E, E, E, E]     # one zero slice for all parameter gradients
        buf = _zeros(sum(sizes), dev)
        parts, off = [], 0
        for n in sizes:
            parts.append(buf[off:off + n])
            off += n
        G = EncoderLayerGrads(**{"d" + n: t.data_ptr() for n, t in zip(_ENC_W_NAMES, parts)})
        w = EncoderLayerWeights(**{n: t.data_ptr() for n, t in zip(_ENC_W_NAMES, params)})
        _EncoderLayerC._images(w, (w_in, w_out, w1, w2), True, B * nq, B * S, E, nq == S)
        dp = C.addressof(d)
        n_ws = lib().hoisdf_encoder_layer_workspace_bytes(dp, 1)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        dx = torch.empty(B, S, E, device=dev)
        gx2 = None if g_x2 is None else g_x2.contiguous()
        gy = None if g_y is None else g_y.contiguous()
        call("hoisdf_encoder_layer_bwd", _p(x), _p(x2), C.addressof(w), dp, _p(saved), saved.numel(), _p(gx2), _p(gy), _p(dx),
             C.addressof(G), _p(ws), n_ws, _st())
        shaped = [parts[0].view(3 * E, E), parts[1], parts[2].view(E, E), parts[3], parts[4], parts[5], parts[6].view(F, E), parts[7],
                  parts[8].view(E, F)] + parts[9:]
        return (dx, None, None, None, *shaped, None, None)


_ENC_W_NAMES = ("w_in", "b_in", "w_out", "b_out", "g1", "be1", "w1", "b1", "w2", "b2", "g2", "be2", "g3", "be3")
_ENCODER_LAYER_C = __import__("os").environ.get("HOISDF_ENCODER_LAYER", "c") != "ops"


def _coarse_layer_ok(p, x, *weights) -> bool:
    """the C entry covers the default arithmetic; the opt-in split / f16 modes and bench.py's per-call event timing (which
    brackets the individual C-ABI calls from Python) take the op-by-op node"""
    from . import _lib
    # (without kept planes the C entry's backward would fall back to the f32 kernel with dQ atomics: in deterministic mode the
    # op-by-op node, whose emulated backward converts on its own, keeps the step order-fixed)
    return (_ENCODER_LAYER_C and _lib._timer is None
            and not _use_f16(p, x, *weights) and not (deterministic() and not _SPLIT_KEEP))


def encoder_layer(x, n_query, p, H, w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3, be3, eps=1e-5,
                  n_inter=None):
    """-> (x2 (B, n_query|S, E) = the layer output, y = inter_norm(x2) (B, n_inter|n_query|S, E): rows < n_inter only)"""
    fn = _EncoderLayerC if _coarse_layer_ok(float(p), x, w_in, w_out, w1, w2) else _EncoderLayer
    return fn.apply(x, n_query, float(p), int(H), w_in, b_in, w_out, b_out, g1, be1, w1, b1, w2, b2, g2, be2, g3,
                               be3, float(eps), n_inter)


class _DecoderLayerC(torch.autograd.Function):
    """common/nets/transformer.py:366-395 (decoder layer, post-norm) + the decoder stack's norm of the layer output through the
    coarse C entries hoisdf_decoder_layer_fwd / _bwd (csrc/layers.hip): one call per direction instead of ~20 / ~45."""

    @staticmethod
    def forward(ctx, tgt, memory, query_pos, mask_u8, kv_len, p, H, eps, *params):
        from ._lib import lib, DecoderLayerDesc, DecoderLayerWeights, _DEC_W
        tgt, memory, query_pos = tgt.contiguous(), memory.contiguous(), query_pos.contiguous()
        _chk(tgt, memory, query_pos, *params)
        assert len(params) == len(_DEC_W) and all(t.is_contiguous() for t in params) and mask_u8.dtype == torch.uint8
        B, Q, E = tgt.shape
        S = memory.shape[1]
        F = params[8].shape[0]
        d = DecoderLayerDesc(B=B, Q=Q, S=S, E=E, F=F, H=H, kv_len=int(kv_len), eps=eps, drop_p=p, training=int(any(ctx.needs_input_grad)))
        for i in range(6):
            d.seed[i] = next_seed() if p > 0 else 0
        w = DecoderLayerWeights(**{n: t.data_ptr() for n, t in zip(_DEC_W, params)})
        ca_w_in = params[4]
        if _GEMM_EMU and B * S >= _GEMM_EMU_MIN_ROWS:
            w.img_ca_kv = _emu_image(ca_w_in[E:], False).data_ptr()
        dp = C.addressof(d)
        n_saved = lib().hoisdf_decoder_layer_saved_bytes(dp) if d.training else 0
        n_ws = lib().hoisdf_decoder_layer_workspace_bytes(dp, 0)
        dev = tgt.device
        saved = torch.empty(n_saved, device=dev, dtype=torch.uint8) if n_saved else None
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        out = torch.empty(B, Q, E, device=dev)
        y = torch.empty(B, Q, E, device=dev)
        call("hoisdf_decoder_layer_fwd", _p(tgt), _p(memory), _p(query_pos), _p(mask_u8), C.addressof(w), dp, _p(out), _p(y), _p(saved),
             n_saved, _p(ws), n_ws, _st())
        ctx.save_for_backward(tgt, memory, mask_u8, out, saved, *params)
        ctx.desc = d
        return out, y

    @staticmethod
    def backward(ctx, g_out, g_y):
        from ._lib import lib, DecoderLayerWeights, DecoderLayerGrads, _DEC_W
        tgt, memory, mask_u8, out, saved, *params = ctx.saved_tensors
        d = ctx.desc
        B, Q, S, E = d.B, d.Q, d.S, d.E
        dev = tgt.device
        sizes = [t.numel() for t in params]
        buf = _zeros(sum(sizes) + Q * E, dev)                     # one zero slice: every parameter gradient + d query_pos
        parts, off = [], 0
        for t, n in zip(params, sizes):
            parts.append(buf[off:off + n].view(t.shape))
            off += n
        d_qpos = buf[off:off + Q * E].view(Q, E)
        G = DecoderLayerGrads(**{"d" + n: t.data_ptr() for n, t in zip(_DEC_W, parts)})
        w = DecoderLayerWeights(**{n: t.data_ptr() for n, t in zip(_DEC_W, params)})
        if _GEMM_EMU and B * S >= _GEMM_EMU_MIN_ROWS:
            w.img_t_ca_kv = _emu_image(params[4][E:], True).data_ptr()
        dp = C.addressof(d)
        n_ws = lib().hoisdf_decoder_layer_workspace_bytes(dp, 1)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        d_tgt = torch.empty(B, Q, E, device=dev)
        d_mem = torch.empty(B, S, E, device=dev)
        go = None if g_out is None else g_out.contiguous()
        gy = None if g_y is None else g_y.contiguous()
        call("hoisdf_decoder_layer_bwd", _p(tgt), _p(memory), _p(mask_u8), _p(out), C.addressof(w), dp, _p(saved), saved.numel(), _p(go), _p(gy),
             _p(d_tgt), _p(d_mem), 0, _p(d_qpos), C.addressof(G), _p(ws), n_ws, _st())
        return (d_tgt, d_mem, d_qpos, None, None, None, None, None, *parts)


_DECODER_LAYER_C = __import__("os").environ.get("HOISDF_DECODER_LAYER", "c") != "ops"


def decoder_layer_ok(p, *tensors) -> bool:
    """as _coarse_layer_ok: default arithmetic only, and not while bench.py brackets the individual calls"""
    from . import _lib
    return (_DECODER_LAYER_C and _lib._timer is None and not _use_f16(p, *tensors))


def decoder_layer(tgt, memory, query_pos, mask_u8, kv_len, p, H, eps, *params):
    """-> (out (B,Q,E), y = the decoder stack's norm of out).  params in the order of _lib._DEC_W (self_attn in / out projection,
    multihead_attn in / out projection, linear1, linear2, norm1..3, stack norm); query_pos (Q,E) is broadcast over the batch."""
    return _DecoderLayerC.apply(tgt, memory, query_pos, mask_u8, int(kv_len), float(p), int(H), float(eps), *params)


# ---------------------------------------------------------------------------------------------
# votes
# ---------------------------------------------------------------------------------------------
class _Vote(torch.autograd.Function):
    @staticmethod
    def forward(ctx, off, cls, pts):
        """off (L,B,P,J*3), cls (L,B,P,J), pts (B,P,3) -> joints (L,B,J,3)"""
        off, cls, pts = off.contiguous(), cls.contiguous(), pts.contiguous()
        _chk(off, cls, pts)
        L, B, P, J = cls.shape
        joints = torch.empty(L, B, J, 3, device=off.device, dtype=torch.float32)
        stats = torch.empty(L, B, J, 2, device=off.device, dtype=torch.float32)
        call("hoisdf_vote_fwd", _p(off), _p(cls), _p(pts), _p(joints), _p(stats), L, B, P, J, _st())
        ctx.save_for_backward(off, cls, pts, joints, stats)
        return joints

    @staticmethod
    def backward(ctx, dj):
        off, cls, pts, joints, stats = ctx.saved_tensors
        L, B, P, J = cls.shape
        dj = dj.contiguous()
        doff = torch.empty_like(off)
        dcls = torch.empty_like(cls)
        call("hoisdf_vote_bwd", _p(off), _p(cls), _p(pts), _p(joints), _p(stats), _p(dj), _p(doff), _p(dcls), L, B, P,
             J, _st())
        return doff, dcls, None


def vote_aggregate(off, cls, pts):
    """K12: joints[l,b,j] = sum_p softmax_p(cls)[p] (pts[p] + off[p,j])."""
    return _Vote.apply(off, cls, pts)


class _VoteLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, off, cls, pts, gt_mm, radius):
        off, cls, pts, gt_mm = off.contiguous(), cls.contiguous(), pts.contiguous(), gt_mm.contiguous()
        _chk(off, cls, pts, gt_mm)
        L, B, P, J = cls.shape
        dev = off.device
        joints = torch.empty(L, B, J, 3, device=dev, dtype=torch.float32)
        stats = torch.empty(L, B, J, 2, device=dev, dtype=torch.float32)
        l3d = torch.empty(L, B, device=dev, dtype=torch.float32)
        bce = torch.empty(L, B, device=dev, dtype=torch.float32)
        near = torch.empty(B, device=dev, dtype=torch.float32)
        call("hoisdf_vote_loss_fwd", _p(off), _p(cls), _p(pts), _p(gt_mm), float(radius), _p(joints), _p(stats), _p(l3d),
             _p(bce), _p(near), L, B, P, J, _st())
        ctx.save_for_backward(off, cls, pts, gt_mm, joints, stats)
        ctx.radius = float(radius)
        ctx.mark_non_differentiable(near)
        return joints, l3d, bce, near

    @staticmethod
    def backward(ctx, dj, dl3d, dbce, _dnear):
        off, cls, pts, gt_mm, joints, stats = ctx.saved_tensors
        L, B, P, J = cls.shape
        doff = torch.empty_like(off)
        dcls = torch.empty_like(cls)
        c = lambda t: None if t is None else t.contiguous()
        dj, dl3d, dbce = c(dj), c(dl3d), c(dbce)
        call("hoisdf_vote_loss_bwd", _p(off), _p(cls), _p(pts), _p(gt_mm), ctx.radius, _p(joints), _p(stats), _p(dj),
             _p(dl3d), _p(dbce), _p(doff), _p(dcls), L, B, P, J, _st())
        return doff, dcls, None, None, None


def vote_loss(off, cls, pts, gt_mm, radius: float):
    """K12 + the JointvoteLoss reductions: -> joints (L,B,J,3), l3d_sum (L,B), bce_sum (L,B), near_sum (B)."""
    return _VoteLoss.apply(off, cls, pts, gt_mm, radius)


# ---------------------------------------------------------------------------------------------
# K1 + K7 + K8 and K11 + K12 as single C calls (include/hoisdf.h hoisdf_tokens_*, hoisdf_heads_vote_*)
# ---------------------------------------------------------------------------------------------
def _mlp_struct(weights, biases, act_last: bool, rows: int = 0, images: str = ""):
    """``images``: "f" -> the cached forward weight images ride along (hoisdf_mlp.img), "b" -> the transposed ones of the grad-input
    GEMMs (img_t), for chains over ``rows`` rows that take the emulated GEMM; the C side builds whatever is missing itself."""
    from ._lib import Mlp, MLP_MAX_LAYERS
    n = len(weights)
    if not 1 <= n <= MLP_MAX_LAYERS:
        raise ValueError(f"an MLP of {n} layers does not fit hoisdf_mlp ({MLP_MAX_LAYERS})")
    m = Mlp()
    m.n_layers, m.act_last = n, int(act_last)
    m.dims[0] = weights[0].shape[1]
    keep = []
    cached = images and _GEMM_EMU and rows >= _GEMM_EMU_MIN_ROWS
    for i, (w, b) in enumerate(zip(weights, biases)):
        if not w.is_contiguous() or w.shape[1] != m.dims[i]:
            raise ValueError("hoisdf_mlp needs dense weights chained dims[i] -> dims[i + 1]")
        m.dims[i + 1] = w.shape[0]
        m.w[i], m.b[i] = w.data_ptr(), b.data_ptr()
        if cached and w.data_ptr() % 16 == 0:
            # forward: contraction = dims[i]; grad-input: contraction = dims[i + 1] (layer 0's input gradient included: hoisdf_tokens_bwd asks for it)
            if images == "f" and w.shape[1] % 4 == 0:
                img = _emu_image(w, False)
                m.img[i] = img.data_ptr()
                keep.append(img)
            elif images == "b" and w.shape[0] % 4 == 0:
                img = _emu_image(w, True)
                m.img_t[i] = img.data_ptr()
                keep.append(img)
    m._keep = keep                  # the images stay alive as long as the struct does
    return m


def _mlp_grads(weights, biases, device):
    """-> (hoisdf_mlp_grads, [dw..., db...] views of ONE zero-filled buffer from the step's arena)"""
    from ._lib import MlpGrads
    sizes = [w.numel() for w in weights] + [b.numel() for b in biases]
    buf = _zeros(sum(sizes), device)
    parts, off = [], 0
    for n in sizes:
        parts.append(buf[off:off + n])
        off += n
    G = MlpGrads()
    k = len(weights)
    for i in range(k):
        G.dw[i], G.db[i] = parts[i].data_ptr(), parts[k + i].data_ptr()
    return G, [parts[i].view_as(weights[i]) for i in range(k)], [parts[k + i] for i in range(k)]


class _Tokens(torch.autograd.Function):
    """hoisdf_tokens_fwd / _bwd on gathered rows: (tok with rows [row0, row0 + P) written, fea)."""

    @staticmethod
    def forward(ctx, tok, feat, cam, center, pe, sdf, beta, row0, n_layers, *wb):
        ws_, bs_ = wb[:n_layers], wb[n_layers:]
        B, S, D = tok.shape
        feat2, cam = _rows(feat), cam.reshape(-1, 3).contiguous()
        P = feat2.shape[0] // B
        pe, sdf = pe.reshape(-1, 30).contiguous(), sdf.reshape(-1).contiguous()
        _chk(tok, feat2, cam, center, pe, sdf, beta, *ws_, *bs_)
        if feat2.stride(0) != feat2.shape[1]:
            feat2 = feat2.contiguous()
        M = B * P
        m = _mlp_struct(ws_, bs_, True, M, "f")
        n_saved = lib().hoisdf_tokens_saved_bytes(C.addressof(m), M, 0)
        n_ws = lib().hoisdf_tokens_workspace_bytes(C.addressof(m), M, 0)
        saved = torch.empty(n_saved, device=tok.device, dtype=torch.uint8)
        ws = torch.empty(n_ws, device=tok.device, dtype=torch.uint8)
        fea = torch.empty(M, D - 33, device=tok.device)
        call("hoisdf_tokens_fwd", None, None, _p(center), None, 1.0, 0, 0, _p(feat2), _p(cam), C.addressof(m), _p(pe), _p(sdf), _p(beta),
             _p(tok), _p(fea), None, B, P, S, row0, D, _p(saved), n_saved, _p(ws), n_ws, _st())
        ctx.save_for_backward(feat2, sdf, beta, saved, *ws_, *bs_)
        ctx.meta = (B, P, S, row0, D, n_layers, feat.shape)
        ctx.mark_dirty(tok)
        ctx.mark_non_differentiable(fea)
        return tok, fea

    @staticmethod
    def backward(ctx, dtok, _dfea):
        feat2, sdf, beta, saved = ctx.saved_tensors[:4]
        B, P, S, row0, D, n, fshape = ctx.meta
        ws_, bs_ = ctx.saved_tensors[4:4 + n], ctx.saved_tensors[4 + n:]
        dev = dtok.device
        dtok = dtok.contiguous()
        M = B * P
        m = _mlp_struct(ws_, bs_, True, M, "b")
        G, dws, dbs = _mlp_grads(ws_, bs_, dev)
        dfeat = torch.empty_like(feat2)
        dbeta = _zeros(1, dev)
        n_ws = lib().hoisdf_tokens_workspace_bytes(C.addressof(m), M, 1)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        call("hoisdf_tokens_bwd", None, None, None, None, 1.0, 0, 0, _p(feat2), C.addressof(m), _p(sdf), _p(beta), _p(dtok), _p(saved),
             saved.numel(), C.addressof(G), _p(dfeat), 0, _p(dbeta), B, P, S, row0, D, _p(ws), n_ws, _st())
        dtok_in = dtok.clone()
        dtok_in[:, row0:row0 + P] = 0            # the rows this op wrote do not depend on the incoming buffer contents
        return (dtok_in, dfeat.view(fshape), None, None, None, None, dbeta, None, None, *dws, *dbs)


def tokens_ok(*tensors) -> bool:
    """the C entry covers the default arithmetic; bench.py's per-call event timing and the split mode take the op chain"""
    from . import _lib
    return _TOKENS_C and _lib._timer is None and all(t.is_cuda for t in tensors)


def tokens(tok, feat, cam, center, pe, sdf, beta, row0: int, weights, biases):
    """K7 + K8 in one C call: rows [row0, row0 + P) of tok (B,S,D) = [cam - center | pe | MLP(feat) * sigma(sdf, beta)];
    -> (tok, fea = MLP(feat) (B P, D - 33), detached: for the cross-field tokens of the other stream)."""
    return _Tokens.apply(tok, feat, cam, center.contiguous(), pe, sdf, beta, row0, len(weights), *weights, *biases)


class _HeadsVote(torch.autograd.Function):
    """hoisdf_heads_vote_fwd / _bwd: enc (L,B,P,E) -> joints (L,B,J,3), l3d_sum (L,B), bce_sum (L,B), near_sum (B)."""

    @staticmethod
    def forward(ctx, enc, pts, gt_mm, radius, nv, nc, *wb):
        vw, vb, cw, cb = wb[:nv], wb[nv:2 * nv], wb[2 * nv:2 * nv + nc], wb[2 * nv + nc:]
        enc, pts, gt_mm = enc.contiguous(), pts.contiguous(), gt_mm.contiguous()
        _chk(enc, pts, gt_mm, *wb)
        L, B, P, E = enc.shape
        J = cw[-1].shape[0]
        dev = enc.device
        mv, mc = _mlp_struct(vw, vb, False, L * B * P, "f"), _mlp_struct(cw, cb, False, L * B * P, "f")
        n_saved = lib().hoisdf_heads_vote_saved_bytes(C.addressof(mv), C.addressof(mc), L, B, P, J)
        n_ws = lib().hoisdf_heads_vote_workspace_bytes(C.addressof(mv), C.addressof(mc), L, B, P, J, 0)
        saved = torch.empty(n_saved, device=dev, dtype=torch.uint8)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        joints = torch.empty(L, B, J, 3, device=dev)
        l3d, bce, near = torch.empty(L, B, device=dev), torch.empty(L, B, device=dev), torch.empty(B, device=dev)
        call("hoisdf_heads_vote_fwd", _p(enc), C.addressof(mv), C.addressof(mc), _p(pts), _p(gt_mm), float(radius), _p(joints), _p(l3d),
             _p(bce), _p(near), L, B, P, J, _p(saved), n_saved, _p(ws), n_ws, _st())
        ctx.save_for_backward(enc, pts, gt_mm, joints, saved, *wb)
        ctx.meta = (float(radius), nv, nc, J)
        ctx.mark_non_differentiable(near)
        return joints, l3d, bce, near

    @staticmethod
    def backward(ctx, dj, dl3d, dbce, _dnear):
        enc, pts, gt_mm, joints, saved = ctx.saved_tensors[:5]
        wb = ctx.saved_tensors[5:]
        radius, nv, nc, J = ctx.meta
        vw, vb, cw, cb = wb[:nv], wb[nv:2 * nv], wb[2 * nv:2 * nv + nc], wb[2 * nv + nc:]
        L, B, P, E = enc.shape
        dev = enc.device
        mv, mc = _mlp_struct(vw, vb, False, L * B * P, "b"), _mlp_struct(cw, cb, False, L * B * P, "b")
        Gv, dvw, dvb = _mlp_grads(vw, vb, dev)
        Gc, dcw, dcb = _mlp_grads(cw, cb, dev)
        n_ws = lib().hoisdf_heads_vote_workspace_bytes(C.addressof(mv), C.addressof(mc), L, B, P, J, 1)
        ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
        denc = torch.empty_like(enc)
        c = lambda t: None if t is None else t.contiguous()
        dj, dl3d, dbce = c(dj), c(dl3d), c(dbce)
        call("hoisdf_heads_vote_bwd", _p(enc), C.addressof(mv), C.addressof(mc), _p(pts), _p(gt_mm), radius, _p(joints), _p(saved),
             saved.numel(), _p(dj), _p(dl3d), _p(dbce), C.addressof(Gv), C.addressof(Gc), _p(denc), L, B, P, J, _p(ws), n_ws, _st())
        return (denc, None, None, None, None, None, *dvw, *dvb, *dcw, *dcb)


def heads_vote(enc, pts, gt_mm, radius: float, vote_w, vote_b, cls_w, cls_b):
    """K11 + K12 in one C call (linear_handvote + linear_handcls on all depths + vote aggregation + JointvoteLoss sums)."""
    return _HeadsVote.apply(enc, pts, gt_mm, radius, len(vote_w), len(cls_w), *vote_w, *vote_b, *cls_w, *cls_b)


class _PointLoss(torch.autograd.Function):
    """a15: hoisdf_point_loss_fwd / _bwd (include/hoisdf.h)."""

    @staticmethod
    def forward(ctx, pred, target, rep, C_, Bt, kind, clamp, pred_scale):
        pred, target = pred.contiguous(), target.contiguous()
        _chk(pred, target)
        n = pred.numel()
        if target.numel() != Bt * C_ or n % (rep * C_) or (n // (rep * C_)) % Bt:
            raise ValueError(f"point_loss: pred {tuple(pred.shape)} is not a broadcast of a ({Bt}, {C_}) target with {rep} repeats")
        part = torch.empty(lib().hoisdf_point_loss_blocks(n), device=pred.device, dtype=torch.float32)
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        call("hoisdf_point_loss_fwd", _p(pred), _p(target), n, rep, C_, Bt, kind, float(clamp), float(pred_scale), 1.0 / n,
             _p(part), _p(loss), _st())
        ctx.save_for_backward(pred, target)
        ctx.args = (n, rep, C_, Bt, kind, float(clamp), float(pred_scale))
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        n, rep, C_, Bt, kind, clamp, pred_scale = ctx.args
        d = torch.empty_like(pred)
        call("hoisdf_point_loss_bwd", _p(pred), _p(target), n, rep, C_, Bt, kind, clamp, pred_scale, 1.0 / n, _p(g.contiguous()),
             _p(d), _st())
        return d, None, None, None, None, None, None, None


def l1_loss_clamped_target(pred, target, clamp: float):
    """mean |pred - clamp(target, +-clamp)| (SepSDFLoss on the clamped ground truth, common/nets/loss.py:64-78 with
    main/model.py:393-400); pred and target hold the same number of elements."""
    return _PointLoss.apply(pred, target, 1, 1, pred.numel(), 0, clamp, 1.0)


def smooth_l1_loss_broadcast(pred, target, rep: int = 1, pred_scale: float = 1.0):
    """torch.nn.SmoothL1Loss()(pred * pred_scale, target expanded): pred (..., Bt, rep, C) against target (Bt, C)
    (main/model.py:656-662: rep = points; common/nets/loss.py:57-59: rep = 1, C = J * 3)."""
    Bt = target.shape[0]
    return _PointLoss.apply(pred, target, rep, target.numel() // Bt, Bt, 1, 0.0, pred_scale)


# ---------------------------------------------------------------------------------------------
# (f3) MANO head: 6D pose -> rotations -> MANO layer -> vertices / joints (+ the four ManoLoss sums), one kernel each way
# ---------------------------------------------------------------------------------------------
def mano_dirs_image(shapedirs, posedirs, weights):
    """th_shapedirs (778,3,10) + th_posedirs (778,3,135) + th_weights (778,16) -> the transposed table image the kernels read
    ([145][2334] directions, then [16][778] skinning weights)."""
    shapedirs, posedirs, weights = shapedirs.contiguous().float(), posedirs.contiguous().float(), weights.contiguous().float()
    _chk(shapedirs, posedirs, weights)
    assert shapedirs.shape == (778, 3, 10) and posedirs.shape == (778, 3, 135) and weights.shape == (778, 16)
    from ._lib import lib
    image = torch.empty(lib().hoisdf_mano_dirs_image_floats(), device=shapedirs.device, dtype=torch.float32)
    call("hoisdf_mano_prepare", _p(shapedirs), _p(posedirs), _p(weights), _p(image), _st())
    return image


def _mano_gt_args(gt):
    if gt is None:
        return (None, None, None, None, 0, 0)
    gv, gj, gr, gshape = gt                       # gshape: (Bg, >= 10) view with unit inner stride (mano_param[:, 48:])
    assert gshape.stride(1) == 1 and gv.is_contiguous() and gj.is_contiguous() and gr.is_contiguous()
    return (_p(gv), _p(gj), _p(gr), _p(gshape), gshape.stride(0), gv.shape[0])


class _ManoHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose6d, betas, assets, gt):
        pose6d, betas = pose6d.contiguous(), betas.contiguous()
        _chk(pose6d, betas)
        H = pose6d.shape[0]
        assert pose6d.shape == (H, 16, 6) and betas.shape == (H, 10)
        dev = pose6d.device
        verts = torch.empty(H, 778, 3, device=dev, dtype=torch.float32)
        joints = torch.empty(H, 21, 3, device=dev, dtype=torch.float32)
        rot = torch.empty(H, 16, 3, 3, device=dev, dtype=torch.float32)
        sums = torch.empty(H, 4, device=dev, dtype=torch.float32) if gt is not None else None
        image, tmpl, jreg, w, mean = assets
        call("hoisdf_mano_head_fwd", _p(pose6d), 96, 0, _p(betas), 10, H, _p(image), _p(tmpl), _p(jreg), _p(w), _p(mean),
             *_mano_gt_args(gt), _p(verts), _p(joints), _p(rot), _p(sums), _st())
        ctx.save_for_backward(pose6d, betas)
        ctx.assets, ctx.gt = assets, gt
        ctx.set_materialize_grads(False)
        return verts, joints, rot, sums             # sums is None without ground truth

    @staticmethod
    def backward(ctx, g_verts, g_joints, g_rot, g_sums):
        pose6d, betas = ctx.saved_tensors
        H = pose6d.shape[0]
        gs = [None if g is None else g.contiguous() for g in (g_sums if ctx.gt is not None else None, g_verts, g_joints, g_rot)]
        d_pose, d_betas = torch.empty_like(pose6d), torch.empty_like(betas)
        image, tmpl, jreg, w, mean = ctx.assets
        call("hoisdf_mano_head_bwd", _p(pose6d), _p(betas), H, _p(image), _p(tmpl), _p(jreg), _p(w), _p(mean),
             *_mano_gt_args(ctx.gt), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(gs[3]), _p(d_pose), _p(d_betas), _st())
        return d_pose, d_betas, None, None


def mano_head(pose6d, betas, assets, gt=None):
    """(f3) mano_head.py:232-250 + manolayer.py:111-276 in one launch: pose6d (H,16,6), betas (H,10) -> verts (H,778,3) and
    joints (H,21,3) in metres (wrist-centred), rot (H,16,3,3) = the Gram-Schmidt rotations, and, with
    gt = (gt_verts, gt_joints, gt_rot, gt_shape) of Bg hands (hand h pairs with h % Bg), sums (H,4) = squared-error sums of
    (verts, joints, rot, betas).  assets = (dirs_image, v_template, J_regressor, weights, hands_mean) - hands_mean must be zero."""
    return _ManoHead.apply(pose6d, betas, assets, gt)


def mano_gt(mano_param, assets):
    """the ground-truth hands of mano_head.py:252-276: mano_param (B, 58) = 48 axis-angle coefficients + 10 betas ->
    verts (B,778,3), joints (B,21,3) in metres, rot (B,16,3,3).  No gradient."""
    mano_param = mano_param.contiguous().float()
    _chk(mano_param)
    B = mano_param.shape[0]
    assert mano_param.shape[1] == 58
    dev = mano_param.device
    verts = torch.empty(B, 778, 3, device=dev, dtype=torch.float32)
    joints = torch.empty(B, 21, 3, device=dev, dtype=torch.float32)
    rot = torch.empty(B, 16, 3, 3, device=dev, dtype=torch.float32)
    image, tmpl, jreg, w, mean = assets
    betas = mano_param[:, 48:]
    call("hoisdf_mano_head_fwd", _p(mano_param), 58, 1, _p(betas), 58, B, _p(image), _p(tmpl), _p(jreg), _p(w), _p(mean),
         None, None, None, None, 0, 0, _p(verts), _p(joints), _p(rot), None, _st())
    return verts, joints, rot


# ---------------------------------------------------------------------------------------------
# (f4) auxiliary image losses of the encoder outputs
# ---------------------------------------------------------------------------------------------
class _AuxImageLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, joints, hand_seg, obj_seg, sigma):
        if joints.dim() != 3 or joints.shape[2] != 2:
            raise RuntimeError(f"aux_image_losses: joint_coord must be (B, J, 2) heat-map pixels, got {tuple(joints.shape)}")
        joints, hand_seg, obj_seg = joints.contiguous().float(), hand_seg.contiguous().float(), obj_seg.contiguous().float()
        _chk(dec, joints, hand_seg, obj_seg)
        B, C, H, W = dec.shape
        assert C >= 3 and hand_seg.shape == (B, H, W) and obj_seg.shape == (B, H, W) and joints.shape[0] == B
        hm, l_hm, l_obj, l_hand = (torch.empty(B, H, W, device=dec.device, dtype=torch.float32) for _ in range(4))
        sb, sc, sh, sw = dec.stride()
        call("hoisdf_aux_image_losses_fwd", _p(dec), sb, sc, sh, sw, _p(joints), _p(hand_seg), _p(obj_seg), B,
             joints.shape[1], H, W, float(sigma), _p(hm), _p(l_hm), _p(l_obj), _p(l_hand), _st())
        ctx.save_for_backward(dec, hand_seg, obj_seg, hm)
        ctx.mark_non_differentiable(hm)
        return l_hm, l_obj, l_hand, hm

    @staticmethod
    def backward(ctx, g_hm, g_obj, g_hand, _g):
        dec, hand_seg, obj_seg, hm = ctx.saved_tensors
        B, C, H, W = dec.shape
        ddec = torch.empty_like(dec) if C == 3 else torch.zeros_like(dec)       # preserves the (channels_last) strides
        assert ddec.stride() == dec.stride()
        gs = [None if g is None else g.contiguous() for g in (g_hm, g_obj, g_hand)]
        sb, sc, sh, sw = dec.stride()
        call("hoisdf_aux_image_losses_bwd", _p(dec), sb, sc, sh, sw, _p(hand_seg), _p(obj_seg), _p(hm), _p(gs[0]),
             _p(gs[1]), _p(gs[2]), B, H, W, _p(ddec), _st())
        return ddec, None, None, None, None


def aux_image_losses(decoder_out, joint_coord, hand_seg, obj_seg, sigma: float):
    """(f4) main/model.py:404-422 in one pass: returns (joint_heatmap, obj_seg, hand_seg) per-pixel losses (B,H,W)
    and the rendered target heat-map; differentiable w.r.t. decoder_out (any strides)."""
    return _AuxImageLosses.apply(decoder_out, joint_coord, hand_seg, obj_seg, sigma)




# ---------------------------------------------------------------------------------------------
# (f4) BatchNorm2d (+ residual) (+ ReLU) of a channels_last encoder map
# ---------------------------------------------------------------------------------------------
def _rows_cl(t: torch.Tensor):
    """a (N, C, H, W) map whose memory is channels_last rows -> (t, row stride) with t usable as [N H W][C] (a channel slice of a
    concatenation keeps its parent's row stride); anything else is made channels_last-dense first"""
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ld = sw if w > 1 else sh if h > 1 else sn if n > 1 else c          # (the stride of a size-1 dimension says nothing)
    ok = ((sc == 1 or c == 1) and ld >= c and ld % 4 == 0 and t.data_ptr() % 16 == 0 and (w == 1 or sw == ld)
          and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld))
    if not ok:
        t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = c
    return t, ld


_BN_WS = {}


def _bn_workspace_floats(M: int, c: int) -> int:
    n = _BN_WS.get((M, c))
    if n is None:
        from ._lib import lib
        n = _BN_WS[(M, c)] = lib().hoisdf_bn_workspace_floats(M, c)
    return n


def bn_act_supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 8 == 0 and 8 <= x.shape[1] <= 2048 and x.numel() > 0


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu):
        _chk(x, weight, bias, residual)
        n, c, h, w = x.shape
        M = n * h * w
        x, ldx = _rows_cl(x)
        ldr = 0
        if residual is not None:
            assert residual.shape == x.shape, (residual.shape, x.shape)
            residual, ldr = _rows_cl(residual)
        y = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)      # channels_last, dense rows
        need_grad = training and (x.requires_grad or (weight is not None and weight.requires_grad) or
                                  (residual is not None and residual.requires_grad))
        if training:
            nws = _bn_workspace_floats(M, c)
            ws = torch.empty(nws, device=x.device, dtype=torch.float32)
            stats = torch.empty(2, c, device=x.device, dtype=torch.float32)
            mean, invstd = stats[0], stats[1]
            call("hoisdf_bn_stats", _p(x), ldx, M, c, _p(mean), _p(invstd), _p(running_mean), _p(running_var),
                 float(momentum if momentum is not None else 0.0), float(eps), _p(ws), nws, _st())
            bits = torch.empty(M, c // 8, device=x.device, dtype=torch.uint8) if (relu and need_grad) else None
            call("hoisdf_bn_apply_fwd", _p(x), ldx, _p(residual), ldr, _p(mean), _p(invstd), 0, float(eps), _p(weight), _p(bias), int(relu),
                 _p(y), _p(bits), M, c, _st())
            ctx.save_for_backward(x, weight, mean, invstd, bits)
            ctx.meta = (M, c, ldx, residual is not None, bool(relu), weight is not None, bias is not None)
        else:
            call("hoisdf_bn_apply_fwd", _p(x), ldx, _p(residual), ldr, _p(running_mean), _p(running_var), 1, float(eps), _p(weight), _p(bias),
                 int(relu), _p(y), None, M, c, _st())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, invstd, bits = ctx.saved_tensors
        M, c, ldx, has_res, relu, has_w, has_b = ctx.meta
        dy, lddy = _rows_cl(dy)
        n, _, h, w = x.shape
        dx = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
        dres = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2) if has_res else None
        dwb = torch.empty(2, c, device=x.device, dtype=torch.float32)
        nws = _bn_workspace_floats(M, c)
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        call("hoisdf_bn_bwd", _p(dy), lddy, _p(x), ldx, _p(bits if relu else None), _p(mean), _p(invstd), _p(weight), _p(dx), _p(dres),
             _p(dwb[0]), _p(dwb[1]), M, c, _p(ws), nws, _st())
        return dx, (dwb[0] if has_w else None), (dwb[1] if has_b else None), dres, None, None, None, None, None, None


def bn_act(x, weight, bias, running_mean, running_var, training: bool, momentum, eps: float, relu: bool, residual=None):
    """(f4) relu?(batch_norm(x) (+ residual)) of a channels_last map in two HBM passes per direction (csrc/bnact.hip);
    training: batch statistics, running statistics updated in place as torch.nn.functional.batch_norm does;
    otherwise the running statistics normalise (no gradient path: evaluation)."""
    return _BnAct.apply(x, weight, bias, residual, running_mean, running_var, bool(training), momentum, float(eps), bool(relu))
