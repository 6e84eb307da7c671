"""CPU tests of everything above the kernels: the C ABI surface (library loads, exports exactly
what include/daam_hip.h declares, fails loudly without a GPU), the hooking framework, the locator,
token merge indices, the engine's deferred-tap bookkeeping (against a recording fake of the native
library -- test-only, the product has no fallback), and the torch port used for the timed baselines."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_pipe, load_golden
from oracle import fake_diffusers as fd
from oracle import heatmap_oracle as ho


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'daam_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(daam_[a-z_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from daam_amd import _native
    lib = _native.load()
    declared = _header_symbols()
    assert declared == sorted(_native.EXPORTS), 'include/daam_hip.h and daam_amd/_native.py disagree'
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.daam_abi_version() == _native.ABI_VERSION == 6
    assert ctypes.sizeof(_native.QKDesc) == 80          # 8 x 4 bytes + 6 x 8 bytes, no padding surprises
    # built with -fvisibility=hidden: the only FUNCTIONS the library exports are the C ABI (the remaining dynamic
    # symbols are the device-kernel handles the HIP runtime registers)
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    functions = sorted(line.split()[-1] for line in nm.splitlines() if line.split()[-2] in ('T', 't'))
    assert functions == declared, set(functions) ^ set(declared)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_context_creation_fails_loudly_without_gpu():
    from daam_amd import _native
    lib = _native.load()
    ctx = ctypes.c_void_p()
    rc = lib.daam_ctx_create(4, 77, 64, 0, ctypes.byref(ctx))
    assert rc != 0 and not ctx.value
    with pytest.raises(_native.DaamError):
        _native.check(rc)
    assert lib.daam_ctx_create(4, 99, 64, 0, ctypes.byref(ctx)) == -1        # DAAM_E_INVALID: tokens > 80
    assert b'tokens' in lib.daam_last_error()
    assert lib.daam_tap_flush(None, None) == -1


# ------------------------------------------------------------------------------------------------
# hook framework + locator (reference hook.py)
# ------------------------------------------------------------------------------------------------
def test_object_hooker_semantics():
    from daam_amd.hook import AggregateHooker, ObjectHooker

    class Thing:
        def f(self, x):
            return x + 1

    class H(ObjectHooker):
        def _hooked_f(hk, thing, x):
            return 10 * hk.monkey_super('f', x)

        def _hook_impl(self):
            self.monkey_patch('f', self._hooked_f)
            self.monkey_patch('missing', self._hooked_f, strict=False)

    t = Thing()
    h = H(t)
    with pytest.raises(RuntimeError, match='Module is not hooked'):
        h.unhook()
    with h:
        assert t.f(1) == 20
        with pytest.raises(RuntimeError, match='Already hooked module'):
            h.hook()
    assert t.f(1) == 2 and not h.hooked
    with pytest.raises(AttributeError):
        h.monkey_patch('missing', None)
    agg = AggregateHooker([H(Thing()), H(Thing())])
    agg.register_hook(H(Thing()))
    with agg:
        assert all(x.hooked for x in agg.module)
    assert not any(x.hooked for x in agg.module)


@pytest.mark.parametrize('kind,n,n_mid', [('sd15', 15, 16), ('sdxl', 60, 70)])
def test_locator_order_matches_oracle(kind, n, n_mid):
    from daam_amd.hook import UNetCrossAttentionLocator
    unet = fd.FakeUNet(kind, mini=True)
    for restrict, mid in [(None, False), (None, True), ({0}, False), ({0, 2}, True)]:
        loc = UNetCrossAttentionLocator(restrict=restrict, locate_middle_block=mid)
        got = loc.locate(unet)
        want, names = ho.locate(unet, restrict, mid)
        assert [id(m) for m in got] == [id(m) for m in want]
        assert loc.layer_names == names
    assert len(UNetCrossAttentionLocator().locate(unet)) == n
    assert len(UNetCrossAttentionLocator(locate_middle_block=True).locate(unet)) == n_mid


def test_layer_names_match_reference_golden(golden_case):
    from daam_amd.hook import UNetCrossAttentionLocator
    name, z, meta = golden_case
    pipe = golden_pipe(meta)
    loc = UNetCrossAttentionLocator(locate_middle_block=bool(meta.get('heads')))        # trace.py:34-35
    loc.locate(pipe.unet)
    assert loc.layer_names == json.loads(str(z['layer_names']))


def test_token_merge_indices():
    from daam_amd.utils import compute_token_merge_indices
    tok = fd.FakeTokenizer()
    prompt = 'A photo of a Monkey riding a bicycle and a monkey'
    assert compute_token_merge_indices(tok, prompt, 'monkey') == ([5, 12], None)
    assert compute_token_merge_indices(tok, prompt, 'bicycle') == ([8, 9], None)      # two sub-word pieces
    assert compute_token_merge_indices(tok, prompt, 'bicycle', offset_idx=1) == ([9, 10], None)
    assert compute_token_merge_indices(tok, prompt, 'x', word_idx=3) == ([4], 3)
    with pytest.raises(ValueError, match='Search word cat not found in prompt!'):
        compute_token_merge_indices(tok, prompt, 'Cat')
    toks = tok.tokenize(prompt.lower())
    assert ho.token_merge_indices(toks, tok.tokenize('bicycle'), 'bicycle')[0] == [8, 9]


# ------------------------------------------------------------------------------------------------
# trace on a CPU pipeline: hooks install / restore; the tap refuses CPU tensors
# ------------------------------------------------------------------------------------------------
def test_trace_installs_and_restores_processors():
    import daam_amd
    z, meta = load_golden('sd15_f32')
    pipe = golden_pipe(meta)
    attn = [s.module for s in pipe.unet.execution_order()]
    before = [a.processor for a in attn]
    orig_check = pipe.check_inputs
    tc = daam_amd.trace(pipe, low_memory=True)
    assert tc.latent_hw == 4096 and len(tc.layer_names) == 6           # restrict={0}: one per block
    tc = daam_amd.trace(pipe)
    with tc:
        hooked = [a for a in attn if type(a.processor).__name__ == 'UNetCrossAttentionHooker']
        assert len(hooked) == 15                                        # mid block not hooked without save/load heads
        assert [a.processor.layer_idx for a in ho.locate(pipe.unet)[0]] == list(range(15))
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            pipe('a dog', num_inference_steps=1)
        assert tc.last_prompt == 'a dog'
        with pytest.raises(RuntimeError, match='Did you forget'):
            tc.compute_global_heat_map()
    assert [a.processor for a in attn] == before
    assert pipe.check_inputs == orig_check
    assert daam_amd.trace is daam_amd.DiffusionHeatMapHooker
    pipe768 = fd.make_pipe('sd15', sample_size=96, mini=True)          # SD-2.x 768: 96x96 latents
    assert daam_amd.trace(pipe768).latent_hw == 9216
    with pytest.raises(ValueError):
        daam_amd.trace(pipe, tap='nope')


# ------------------------------------------------------------------------------------------------
# engine bookkeeping against a recording fake of libdaam_hip (test-only)
# ------------------------------------------------------------------------------------------------
class _FakeLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith('daam_'):
            raise AttributeError(name)

        def fn(*args):
            self.calls.append((name, args))
            if name == 'daam_ctx_create':
                args[-1]._obj.value = 1234
            if name == 'daam_key_offset':
                return 0
            return 0
        return fn

    def names(self):
        return [c[0] for c in self.calls]


@pytest.fixture
def fake_engine(monkeypatch):
    from daam_amd import engine as E
    lib = _FakeLib()
    monkeypatch.setattr(E.nat, 'load', lambda: lib)
    monkeypatch.setattr(E.HeatMapEngine, '_require_device',
                        lambda self, t: setattr(self, 'device', torch.device('cpu')))
    class _Stream:
        cuda_stream = 0

        def wait_stream(self, other):
            pass

        def wait_event(self, ev):
            pass

        def record_event(self):
            return object()
    one = _Stream()
    monkeypatch.setattr(E.HeatMapEngine, '_current_stream', lambda self: one)
    monkeypatch.setattr(torch.cuda, 'device', lambda d: __import__('contextlib').nullcontext())
    return E, lib


@pytest.mark.parametrize('recorder', ['c++', 'python'])
def test_engine_deferred_bookkeeping(fake_engine, monkeypatch, recorder):
    E, lib = fake_engine
    if recorder == 'python':
        monkeypatch.setenv('DAAM_NO_FASTPATH', '1')
    eng = E.HeatMapEngine(3, defer_steps=2)
    assert (eng._fast is not None) == (recorder == 'c++')    # build() compiles daam_amd._fastpath
    q = [torch.zeros(2, 64, 16, dtype=torch.float16) for _ in range(3)]
    k = [torch.zeros(2, 77, 16, dtype=torch.float16) for _ in range(3)]
    for step in range(5):
        for layer in (2, 0, 1):                       # execution order != locator order
            eng.tap_qk(layer, q[layer], k[layer], heads=2, scale=0.35, factor=8 // 8 or 1)
    # 5 steps at 2 per launch: flushed after
    # steps 2 and 4 (when step 3 / 5 arrive), 3 taps still recorded
    assert lib.names().count('daam_tap_flush') == 2
    many = [c for c in lib.calls if c[0] == 'daam_tap_qk_enqueue_many']
    assert [c[1][1] for c in many] == [6, 6]
    assert eng.pending_taps == 3 and eng.touched == [2, 0, 1]
    assert eng.keys()[:2] == [(1, 2, 0), (1, 2, 1)]
    list(eng.items())                                  # reading the sums flushes the rest
    assert lib.names().count('daam_tap_flush') == 3 and not eng.pending_taps
    assert lib.names().count('daam_layer_configure') == 3
    # a shape change of a layer mid-batch starts a new batch and a new buffer
    eng.tap_qk(0, q[0], k[0], 2, 0.35, 1)
    eng.tap_qk(0, torch.zeros(2, 256, 16, dtype=torch.float16), k[0], 2, 0.35, 1)
    assert lib.names().count('daam_tap_flush') == 4 and lib.names().count('daam_layer_configure') == 4
    eng.clear()                                        # RawHeatMapCollection.clear: drop recorded taps, zero sums
    # the sums were handed out as views above (items()): after the reset the context forgets every buffer (daam_layer_release)
    # so that nothing native ever writes into memory the views own
    tail = lib.names()[lib.names().index('daam_reset', len(lib.names()) - 5):]
    assert not eng.pending_taps and not eng.touched and tail == ['daam_reset'] + ['daam_layer_release'] * 3
    with pytest.raises(LookupError):
        eng.global_heat_map()
    eng.close()
    assert lib.names()[-1] == 'daam_ctx_destroy'


def test_cxx_recorder_steady_state(fake_engine):
    """daam_amd._fastpath: positional steady-state calls never re-enter Python; anything unusual does."""
    import ctypes
    E, lib = fake_engine
    eng = E.HeatMapEngine(2, defer_steps=3)
    assert eng._fast is not None and eng.tap_qk == eng._fast.tap
    slow = []
    orig = eng._prepare_qk
    eng._prepare_qk = lambda *a: (slow.append(a[0]), orig(*a))[1]
    q = [torch.zeros(2, 64, 16, dtype=torch.float16) for _ in range(4)]
    k = [torch.zeros(2, 77, 16, dtype=torch.float16) for _ in range(4)]
    for step in range(3):
        for layer in (1, 0):
            eng.tap_qk(layer, q[2 * layer + step % 2], k[2 * layer + step % 2], 2, 0.35, 1)
    assert slow == [1, 0] and eng.pending_taps == 6 and 'daam_tap_flush' not in lib.names()
    n, la, qa, ka, da = eng._fast.buffers()
    assert n == 6
    assert list((ctypes.c_int32 * n).from_address(la)) == [1, 0, 1, 0, 1, 0]
    assert list((ctypes.c_uint64 * n).from_address(qa)) == [q[i].data_ptr() for i in (2, 0, 3, 1, 2, 0)]
    assert list((ctypes.c_uint64 * n).from_address(ka)) == [k[i].data_ptr() for i in (2, 0, 3, 1, 2, 0)]
    descs = list((ctypes.c_uint64 * n).from_address(da))
    assert descs[0] == eng._qk_cache[1].desc_addr and descs[1] == eng._qk_cache[0].desc_addr
    # the window (3 steps) is full: the next call launches first
    eng.tap_qk(1, q[2], k[2], 2, 0.35, 1)
    assert lib.names().count('daam_tap_flush') == 1 and eng.pending_taps == 1 and slow == [1, 0]
    # changed call parameters, a non-contiguous input, keyword arguments: all take the Python path
    eng.tap_qk(0, q[0], k[0], 2, 0.5, 1)                                   # scale changed
    assert slow == [1, 0, 0]
    eng.tap_qk(1, torch.zeros(2, 16, 64, dtype=torch.float16).transpose(1, 2), k[2], 2, 0.35, 1)
    assert slow == [1, 0, 0, 1]
    eng.tap_qk(0, q[0], k[0], heads=2, scale=0.5, factor=1)
    assert slow == [1, 0, 0, 1]                                            # cache hit on the Python side
    with pytest.raises(IndexError):
        eng.tap_qk(7, q[0], k[0], 2, 0.35, 1)
    with pytest.raises(TypeError):
        eng.tap_qk(0, q[0])
    # the recorder keeps Q / K alive until the launch, then lets go
    t = torch.zeros(2, 64, 16, dtype=torch.float16)
    eng.tap_qk(0, t, k[0], 2, 0.5, 1)
    del t
    eng.flush()
    assert eng.pending_taps == 0
    eng.clear()
    assert eng._fast.get_window() == 3 and eng.touched == []
    eng.close()


def test_cxx_recorder_releases_storages_on_its_thread(fake_engine):
    """The recorder keeps the STORAGE of a recorded Q / K (the tensor object dies with the processor call, as it would
    without a trace) and hands the storages of a launch to its release thread; ``drain_released`` waits for it."""
    import weakref
    E, lib = fake_engine
    from daam_amd import _fastpath
    eng = E.HeatMapEngine(1, defer_steps=64)
    k = torch.zeros(2, 77, 16, dtype=torch.float16)
    watched = []
    for step in range(40):                                   # > 64 storages: the batch goes to the thread
        q = torch.zeros(2, 64, 16, dtype=torch.float16)
        if step % 8 == 0:
            watched.append((weakref.ref(q), q.untyped_storage()))     # a storage with a Python object: freed under the GIL
        eng.tap_qk(0, q, k, 2, 0.35, 1)
        del q
    assert all(ref() is None for ref, _ in watched)          # tensors are gone, their memory is held
    held = [torch._C._storage_Use_Count(st._cdata) for _, st in watched]
    eng.flush()
    E.drain_released()
    assert [torch._C._storage_Use_Count(st._cdata) for _, st in watched] == [c - 1 for c in held]
    assert eng.pending_taps == 0 and eng._fast.held_bytes() == 0
    # inline release gives the same result
    _fastpath.set_sync_release(True)
    try:
        q = torch.zeros(2, 64, 16, dtype=torch.float16)
        st = q.untyped_storage()
        for _ in range(70):
            eng.tap_qk(0, q, k, 2, 0.35, 1)
            if eng._fast.pending(0) == 64:
                break
        before = torch._C._storage_Use_Count(st._cdata)
        eng.flush()
        assert torch._C._storage_Use_Count(st._cdata) == before - 64
    finally:
        _fastpath.set_sync_release(False)
    eng.close()


def test_engine_attend_bookkeeping(fake_engine):
    """``HeatMapEngine.attend`` against the recording fake of libdaam_hip: which calls take the kernel, what the
    descriptor says, fused tap on an immediate trace vs recorded tap on a deferred one."""
    import ctypes
    E, lib = fake_engine
    q = torch.zeros(2, 64, 4 * 40, dtype=torch.float16)                    # 4 heads of 40 (SD-v1.5's 64x64 layers)
    k, v = torch.zeros(2, 77, 160, dtype=torch.float16), torch.zeros(2, 77, 160, dtype=torch.float16)
    # deferred trace: the kernel attends (tap = 0) and the call is recorded for the batched launch
    eng = E.HeatMapEngine(2, defer_steps=4)
    out = eng.attend(1, q, k, v, 4, 40 ** -0.5, 1, True, tapped=True)
    assert out is not None and out.shape == q.shape and out.dtype == q.dtype
    name, args = lib.calls[-1] if lib.calls[-1][0] == 'daam_attend' else [c for c in lib.calls if c[0] == 'daam_attend'][-1]
    assert args[1] == 1 and args[7] == 0
    assert args[2:6] == (q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr())
    desc = ctypes.cast(args[6], ctypes.POINTER(E.nat.AttendDesc)).contents if not hasattr(args[6], '_obj') else args[6]._obj
    assert (desc.qk.batch, desc.qk.heads, desc.qk.hw, desc.qk.tokens, desc.qk.head_dim) == (2, 4, 64, 77, 40)
    assert (desc.qk.q_stride_b, desc.qk.q_stride_h, desc.qk.q_stride_p) == (64 * 160, 40, 160)
    assert (desc.v_stride_b, desc.v_stride_h, desc.v_stride_t) == (77 * 160, 40, 160)
    assert (desc.o_stride_b, desc.o_stride_h, desc.o_stride_p) == (64 * 160, 40, 160)
    assert eng.pending_taps == 1 and eng.touched == [1] and 'daam_tap_qk' not in lib.names()
    # an un-tapped call (reference gate, trace.py:289) attends only
    eng.attend(0, q, k, v, 4, 40 ** -0.5, 8, True, tapped=False)
    assert eng.pending_taps == 1 and eng.touched == [1]
    # immediate trace: the tap is fused into the kernel (tap = 1), nothing is recorded, no stand-alone tap is launched
    imm = E.HeatMapEngine(2, defer_steps=0)
    n_before = len(lib.calls)
    out = imm.attend(0, q, k, v, 4, 40 ** -0.5, 1, True, tapped=True)
    new = [c for c in lib.calls[n_before:]]
    assert [c[0] for c in new if c[0] in ('daam_attend', 'daam_tap_qk', 'daam_layer_configure')] == ['daam_layer_configure', 'daam_attend']
    assert new[-1][1][7] == 1 and imm.pending_taps == 0 and imm.touched == [0]
    assert imm.layer_info[0] == (1, 4, 8)                                   # (factor, kept heads, side)
    # calls the kernel does not take: fp32 pipeline, head_dim not a multiple of 8, not 77 keys, strided input, autograd
    n_before = len(lib.calls)
    assert imm.attend(1, q.float(), k.float(), v.float(), 4, 40 ** -0.5, 1, True, tapped=True) is None
    assert imm.attend(1, q[:, :, :48].contiguous(), k[:, :, :48].contiguous(), v[:, :, :48].contiguous(), 4, 12 ** -0.5, 1) is None
    assert imm.attend(1, q, k[:, :64].contiguous(), v[:, :64].contiguous(), 4, 40 ** -0.5, 1) is None
    assert imm.attend(1, q.transpose(0, 1).contiguous().transpose(0, 1), k, v, 4, 40 ** -0.5, 1) is None
    qg = q.clone().requires_grad_(True)
    assert imm.attend(1, qg, k, v, 4, 40 ** -0.5, 1) is None
    with torch.no_grad():
        assert imm.attend(1, qg, k, v, 4, 40 ** -0.5, 1) is not None
    assert [c[0] for c in lib.calls[n_before:]].count('daam_attend') == 1
    # a library that declines (unaligned view) makes the caller fall back instead of failing
    lib.__dict__['daam_attend'] = lambda *a: E.nat.E_UNSUPPORTED
    assert imm.attend(0, q, k, v, 4, 40 ** -0.5, 1, True, tapped=True) is None
    eng.close()
    imm.close()


@pytest.mark.parametrize('recorder', ['c++', 'python'])
def test_defer_byte_budget(fake_engine, monkeypatch, recorder):
    """The recorded Q / K are kept alive until their launch: a byte budget forces the launch early."""
    E, lib = fake_engine
    if recorder == 'python':
        monkeypatch.setenv('DAAM_NO_FASTPATH', '1')
    q, k = torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16)
    per_tap = (q.numel() + k.numel()) * 2
    eng = E.HeatMapEngine(2, defer_steps=64, defer_bytes=5 * per_tap)
    for step in range(7):
        for layer in (1, 0):
            eng.tap_qk(layer, q, k, 2, 0.35, 1)
    # 2 taps per step; the budget (5 taps) is exceeded after 3 steps, and launches happen on step
    # boundaries only: when steps 4 and 7 begin
    assert lib.names().count('daam_tap_flush') == 2 and eng.pending_taps == 2
    many = [c for c in lib.calls if c[0] == 'daam_tap_qk_enqueue_many']
    assert [c[1][1] for c in many] == [6, 6]
    eng.close()


def test_trace_and_engine_are_freed_without_gc(fake_engine):
    """No reference cycles between trace, hookers, engine and the C++ recorder: dropping the last reference closes
    the native context right away (the running sums of an SDXL trace are 221 MB)."""
    import gc
    import weakref
    import daam_amd
    from oracle import fake_diffusers as fd
    E, lib = fake_engine
    gc.collect()
    gc.disable()
    try:
        eng = E.HeatMapEngine(2, defer_steps=4)
        eng.tap_qk(0, torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16), 2, 0.35, 1)
        ref = weakref.ref(eng)
        del eng
        assert ref() is None and lib.names()[-1] == 'daam_ctx_destroy'
        pipe = fd.make_pipe('sd15', mini=True)
        tc = daam_amd.trace(pipe)
        tref, eref = weakref.ref(tc), weakref.ref(tc.engine)
        with tc:
            pass
        del tc
        assert tref() is None and eref() is None
    finally:
        gc.enable()


def test_context_reuse_between_engines(fake_engine):
    """``reuse_context=True`` (what ``trace`` uses): a closed engine parks its native context + sum buffers, the next
    engine of the same geometry adopts them (reset, no new context); a different geometry gets its own."""
    E, lib = fake_engine
    E._PARKED.clear()
    q, k = torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16)
    a = E.HeatMapEngine(2, defer_steps=4, reuse_context=True)
    a.tap_qk(0, q, k, 2, 0.35, 1)
    buf = a.acc[0]
    a.close()
    assert 'daam_ctx_destroy' not in lib.names() and sum(len(v) for v in E._PARKED.values()) == 1
    b = E.HeatMapEngine(2, defer_steps=4, reuse_context=True)
    n_create = lib.names().count('daam_ctx_create')
    b.tap_qk(0, q, k, 2, 0.35, 1)
    assert lib.names().count('daam_ctx_create') == n_create            # adopted, not created
    assert 'daam_reset' in lib.names() and b.acc[0] is buf and not E._PARKED[b._park_key()]
    assert lib.names().count('daam_layer_configure') == 1               # same layer geometry: buffer kept
    c = E.HeatMapEngine(3, defer_steps=4, reuse_context=True)           # other geometry: its own context
    c.tap_qk(0, q, k, 2, 0.35, 1)
    assert lib.names().count('daam_ctx_create') == n_create + 1
    d = E.HeatMapEngine(2, defer_steps=4)                               # reuse not requested: plain create / destroy
    d.tap_qk(0, q, k, 2, 0.35, 1)
    d.close()
    assert lib.names()[-1] == 'daam_ctx_destroy'
    b.close(); c.close()
    assert sum(len(v) for v in E._PARKED.values()) == 2
    E.release_parked_contexts()
    assert not E._PARKED and lib.names().count('daam_ctx_destroy') == 3


def test_views_keep_their_buffers(fake_engine):
    """The tensors ``all_heat_maps`` hands out are views of the live sums.  Like the reference's tensors they must survive
    ``clear()`` and the end of the trace: an engine whose buffers were handed out starts the next generation on fresh ones
    and is not parked for reuse."""
    E, lib = fake_engine
    E._PARKED.clear()
    q, k = torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16)
    a = E.HeatMapEngine(1, defer_steps=4, reuse_context=True)
    a.tap_qk(0, q, k, 2, 0.35, 1)
    views = [v for _, v in a.items()]
    buf = a.acc[0]
    assert a._views_out and views[0].data_ptr() == buf.data_ptr()
    a.clear()                                                           # next generation: new buffers, the views keep the old ones
    assert not a.acc and not a._views_out and lib.names()[-2:] == ['daam_reset', 'daam_layer_release']
    a.tap_qk(0, q, k, 2, 0.35, 1)
    assert a.acc[0].data_ptr() != buf.data_ptr() and lib.names().count('daam_layer_configure') == 2
    list(a.items())
    a.close()                                                           # views out: destroyed, not parked
    assert lib.names()[-1] == 'daam_ctx_destroy' and not any(E._PARKED.values())
    b = E.HeatMapEngine(1, defer_steps=4, reuse_context=True)           # never iterated: parked as before
    b.tap_qk(0, q, k, 2, 0.35, 1)
    b.close()
    assert sum(len(v) for v in E._PARKED.values()) == 1
    E.release_parked_contexts()


def test_in_place_guard_and_stream_handover(fake_engine, monkeypatch):
    """DAAM_CHECK_VERSIONS=1: the Python recorder remembers tensor._version and refuses to launch after an in-place write;
    a flush from another stream than the one the generation was recorded on orders the two streams both ways."""
    E, lib = fake_engine
    monkeypatch.setenv('DAAM_CHECK_VERSIONS', '1')
    eng = E.HeatMapEngine(1, defer_steps=4)
    assert eng._fast is None                                            # the C++ recorder keeps no versions
    q, k = torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16)
    eng.tap_qk(0, q, k, 2, 0.35, 1)
    eng.flush()
    eng.tap_qk(0, q, k, 2, 0.35, 1)
    q.add_(1)
    with pytest.raises(RuntimeError, match='modified in place'):
        eng.flush()
    assert not eng.pending_taps
    monkeypatch.delenv('DAAM_CHECK_VERSIONS')

    class _S:
        def __init__(self, h):
            self.cuda_stream, self.waited = h, []

        def wait_stream(self, other):
            self.waited.append(other.cuda_stream)

        def wait_event(self, ev):
            pass

        def record_event(self):
            return object()
    rec, cur = _S(11), _S(22)
    eng2 = E.HeatMapEngine(1, defer_steps=4)
    monkeypatch.setattr(E.HeatMapEngine, '_current_stream', lambda self: rec)
    eng2.clear()
    eng2.tap_qk(0, q, k, 2, 0.35, 1)                                    # first tap of the generation: recording stream = 11
    monkeypatch.setattr(E.HeatMapEngine, '_current_stream', lambda self: cur)
    eng2.flush()                                                        # read from stream 22
    assert cur.waited == [11] and rec.waited == [22]
    flush = [c for c in lib.calls if c[0] == 'daam_tap_flush'][-1]
    assert flush[1][1] == 22                                            # the launch goes to the reading stream


def test_bench_refuses_fewer_gpus_than_asked(monkeypatch):
    """``python bench.py --gpus 8`` on a box with fewer devices must not print an n_gpus: 1 line."""
    import bench
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    with pytest.raises(SystemExit, match='only 1 GPU'):
        bench._respawn_under_launcher(8, shared_device=False)


def test_defer_budget_defaults(monkeypatch):
    """$DAAM_DEFER_BYTES wins; otherwise 40 % of the device memory that is free (incl. what torch's caching allocator holds
    unused) when the trace is set up, at least 1 GiB; 32 GiB when there is no device to ask."""
    import sys
    import daam_amd  # noqa: F401
    T = sys.modules['daam_amd.trace']          # `daam_amd.trace` the NAME is the class, like the reference's

    class _P:
        device = torch.device('cuda', 0)

    class _Pipe:
        class unet:
            @staticmethod
            def parameters():
                return iter([_P()])

    monkeypatch.delenv('DAAM_DEFER_BYTES', raising=False)
    monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda dev: 10 << 30)
    monkeypatch.setattr(torch.cuda, 'memory_allocated', lambda dev: 10 << 30)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda dev: (250 << 30, 288 << 30))
    assert T._default_defer_bytes(_Pipe()) == int((250 << 30) * 0.4)          # SDXL-2048: 64 steps x 1.55 GB fit one launch
    assert T._default_defer_bytes(_Pipe()) >= 64 * 1_550_000_000
    monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda dev: 30 << 30)       # 20 GiB cached but unused: as good as free
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda dev: (20 << 30, 288 << 30))
    assert T._default_defer_bytes(_Pipe()) == int((40 << 30) * 0.4)
    monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda dev: 10 << 30)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda dev: (1 << 30, 288 << 30))
    assert T._default_defer_bytes(_Pipe()) == 1 << 30
    assert T._default_defer_bytes(object()) == 32 << 30            # no parameters to ask: the default
    # cpu-offloaded pipeline: parameters on the host, work on the current HIP device -> that device's free memory counts
    _P.device = torch.device('cpu')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda dev: (10 << 30, 288 << 30))
    assert T._default_defer_bytes(_Pipe()) == 4 << 30
    _P.device = torch.device('cuda', 0)
    monkeypatch.setenv('DAAM_DEFER_BYTES', str(5 << 30))
    assert T._default_defer_bytes(_Pipe()) == 5 << 30
    monkeypatch.setenv('DAAM_DEFER_STEPS', '7')
    assert T._default_defer() == 7


def test_engine_immediate_mode_and_dtype_rules(fake_engine):
    E, lib = fake_engine
    eng = E.HeatMapEngine(2, defer_steps=0)
    eng.tap_qk(0, torch.zeros(2, 64, 16, dtype=torch.float16), torch.zeros(2, 77, 16, dtype=torch.float16), 2, 0.35, 1)
    assert 'daam_tap_qk' in lib.names() and 'daam_tap_qk_enqueue_many' not in lib.names()
    assert eng.acc_dtype == torch.float16 and eng.acc[0].shape == (2, 77, 8, 8)
    with pytest.raises(RuntimeError, match='fp32 activations'):
        eng.tap_qk(1, torch.zeros(2, 64, 16), torch.zeros(2, 77, 16), 2, 0.35, 1)
    with pytest.raises(RuntimeError, match='running sums are torch.float16'):      # bf16 into fp16 sums
        eng.tap_qk(1, torch.zeros(2, 64, 16, dtype=torch.bfloat16), torch.zeros(2, 77, 16, dtype=torch.bfloat16), 2, 0.35, 1)
    engb = E.HeatMapEngine(1)
    engb.tap_qk(0, torch.zeros(2, 64, 16, dtype=torch.bfloat16), torch.zeros(2, 77, 16, dtype=torch.bfloat16), 2, 0.35, 1)
    assert engb.acc_dtype == torch.bfloat16 and engb.acc[0].dtype == torch.bfloat16
    with pytest.raises(RuntimeError, match='unsupported pipeline dtype'):
        E.HeatMapEngine(1).tap_qk(0, torch.zeros(2, 64, 16, dtype=torch.float64), torch.zeros(2, 77, 16, dtype=torch.float64), 2, 0.35, 1)
    eng32 = E.HeatMapEngine(1, accumulate='float32')
    eng32.tap_probs(0, torch.zeros(4, 64, 77, dtype=torch.float16), factor=1)
    assert eng32.acc_dtype == torch.float32
    with pytest.raises(ValueError):
        E.HeatMapEngine(1, accumulate='bf16')


# ------------------------------------------------------------------------------------------------
# the torch port (bench baselines) is pinned to the same golden vectors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['sd15_f32', 'sdxl_f32', 'sd15_f16'])
def test_torch_port_matches_reference_golden(name):
    from oracle import torch_hooks as th
    z, meta = load_golden(name)
    pipe = golden_pipe(meta)
    modules, _ = ho.locate(pipe.unet)
    index_of = {id(m): i for i, m in enumerate(modules)}
    lat = ho.latent_hw_for(pipe.unet.config.sample_size, pipe.vae_scale_factor)
    raw = th.RawMaps()
    for step in range(meta['steps']):
        for i, spec in enumerate(pipe.unet.execution_order()):
            li = index_of.get(id(spec.module))
            if li is None:
                continue
            a = spec.module
            q = a.head_to_batch_dim(a.to_q(pipe.hidden_states(i, spec, step)))
            k = a.head_to_batch_dim(a.to_k(pipe.context(i, spec)))
            th.tap(raw, li, th.attention_probs(q, k, a.scale), lat)
    assert np.array_equal(np.asarray([k for k, _ in raw], dtype=np.int32), z['keys'])
    n_rows = len(pipe.tokenizer.tokenize(meta['prompt'])) + 2
    for vn, kw in json.loads(str(z['variants'])).items():
        got = th.global_heat_map(raw, lat, n_rows=n_rows, **kw).numpy()
        tol = 2e-6 if meta['dtype'] == 'float32' else 2e-4
        np.testing.assert_allclose(got, z[f'global_{vn}'], rtol=0, atol=tol * max(1.0, float(np.abs(z[f'global_{vn}']).max())))
    assert len(th.topology('sdxl')) == 60 and sum(h for _, h, _, _ in th.topology('sdxl')) == 1100
    assert len(th.topology('sd15')) == 15 and sum(h for _, h, _, _ in th.topology('sd15')) == 120


@pytest.mark.parametrize('name', ['sd15_f32', 'sdxl_heads_f32', 'sd15_upcast_attn_f16'])
def test_torch_port_processor_matches_reference_golden(name):
    """``oracle.torch_hooks.ReferenceProcessor`` (the reference's processor restated, used as the yardstick of the
    full-size GPU parity tests) returns what the REFERENCE's processor returned and taps the same maps."""
    from oracle import torch_hooks as th
    from oracle.make_golden import OUT_SAMPLE_ROWS
    z, meta = load_golden(name)
    pipe = golden_pipe(meta)
    pipe.keep_outputs = True
    modules, _ = ho.locate(pipe.unet, locate_middle_block=bool(meta.get('heads')))
    raw = th.RawMaps()
    lat = ho.latent_hw_for(pipe.unet.config.sample_size, pipe.vae_scale_factor)
    for idx, m in enumerate(modules):
        m.set_processor(th.ReferenceProcessor(raw, idx, lat))
    pipe(meta['prompt'], num_inference_steps=meta['steps'])
    assert np.array_equal(np.asarray([k for k, _ in raw], dtype=np.int32), z['keys'])
    rel = 1e-6 if meta['dtype'] == 'float32' else 1e-3
    for i, o in enumerate(pipe.last_outputs):
        want = z[f'out_rows_{i}']
        got = o[-1, :OUT_SAMPLE_ROWS].float().numpy()
        assert np.abs(got - want).max() <= rel * max(float(np.abs(want).max()), 1e-6)
        assert abs(float((o.double() ** 2).sum()) - z['out_sums'][i, 1]) <= 4 * rel * z['out_sums'][i, 1]
    n_rows = len(pipe.tokenizer.tokenize(meta['prompt'])) + 2
    got = th.global_heat_map(raw, lat, n_rows=n_rows).numpy()
    np.testing.assert_allclose(got, z['global_default'], rtol=0, atol=(2e-6 if meta['dtype'] == 'float32' else 2e-4) *
                               max(1.0, float(np.abs(z['global_default']).max())))


# ------------------------------------------------------------------------------------------------
# GenerationExperiment: the reference's on-disk layout (experiment.py:140-167, 303-344)
# ------------------------------------------------------------------------------------------------
def test_generation_experiment_round_trip(tmp_path):
    from daam_amd import GenerationExperiment
    import PIL.Image
    img = PIL.Image.fromarray(np.full((8, 8, 3), 200, dtype=np.uint8))
    maps = torch.rand(4, 64, 64)
    exp = GenerationExperiment(img, maps, 'a dog', seed=7, id='p0', path=str(tmp_path), subtype='base',
                               tokenizer=fd.FakeTokenizer())
    assert exp.path == tmp_path / 'p0' and not exp.nsfw()
    exp.annotate('k', [1, 2]).save()
    root = tmp_path / 'p0'
    assert (root / 'prompt.txt').read_text() == 'a dog' and (root / 'seed.txt').read_text() == '7'
    assert (root / 'base' / 'generation.pt').exists() and (root / 'base' / 'output.png').exists()
    assert json.loads((root / 'annotations.json').read_text()) == {'k': [1, 2]}
    back = GenerationExperiment.load(root, subtype='base')
    assert back.prompt == 'a dog' and back.seed == 7 and torch.equal(back.global_heat_map, maps)
    assert back.annotations == {'k': [1, 2]} and back.subtype == 'base' and back.path == root
    assert GenerationExperiment.read_seed(tmp_path, 'p0') == 7 and GenerationExperiment.read_prompt(root) == 'a dog'
    assert back.heat_map().prompt == 'a dog'
    back.clear_checkpoint()
    assert not (root / 'base' / 'generation.pt').exists()


def test_generated_finalize_schedule_is_current(tmp_path):
    """daam_amd/csrc/daam_finalize_pipe_{prefill,asm}_*.inc are generated (tools/gen_fin_pipe.py: one schedule per dtype of the sums -- fp16
    ``r16``, ``bf16`` (shares the fp16 prefill), ``f32``): the committed files must be what the generator writes today."""
    import subprocess
    import sys
    env = dict(os.environ, DAAM_PIPE_OUTDIR=str(tmp_path))
    for k in ('DAAM_PIPE_ABLATE', 'DAAM_PIPE_SCHED', 'DAAM_PIPE_NT', 'DAAM_PIPE_OUT', 'DAAM_PIPE_RING', 'DAAM_PIPE_DT', 'DAAM_PIPE_FAIR'):
        env.pop(k, None)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_fin_pipe.py')], env=env, check=True, capture_output=True)
    names = sorted(f for f in os.listdir(os.path.join(ROOT, 'daam_amd', 'csrc')) if f.startswith('daam_finalize_pipe_') and f.endswith('.inc'))
    assert names == sorted(os.listdir(str(tmp_path))) == ['daam_finalize_pipe_asm_bf16.inc', 'daam_finalize_pipe_asm_f32.inc', 'daam_finalize_pipe_asm_r16.inc',
                                                          'daam_finalize_pipe_prefill_f32.inc', 'daam_finalize_pipe_prefill_r16.inc']
    for name in names:
        assert open(os.path.join(str(tmp_path), name)).read() == open(os.path.join(ROOT, 'daam_amd', 'csrc', name)).read(), name


def test_x2_tap_matrix_as_mfma_operands():
    """What the matrix-core finalize assumes of the 32 -> 64 bicubic table (daam_api.hip: build_up32_ops), restated with the oracle's taps:
    every border-merged weight is an fp16 number (the fp16 / f32 forms take the matrix as an fp16 operand); as bf16 operands it needs
    TWO pieces -- W' = W truncated to eight significant bits and E = W - W' -- because the three taps clamped onto a border column add up
    to 283/256 (nine bits); E is a bf16 number and lives in source columns 0..7 / 24..31 only (the third pass-1 MFMA of the bf16 form reads
    exactly those columns of the plane)."""
    idx, w = ho.bicubic_taps(32, 64)
    W = np.zeros((64, 32), np.float64)
    for o in range(64):
        for a in range(4):
            W[o, idx[o, a]] += float(w[o, a])
    assert np.array_equal(W.astype(np.float16).astype(np.float64), W)
    bits = W.astype(np.float32).view(np.uint32)
    hi = (bits & 0xffff0000).view(np.float32).astype(np.float64)
    lo = W - hi
    assert np.array_equal(ho.round_bf16(lo.astype(np.float32)).astype(np.float64), lo)
    assert sorted(zip(*np.nonzero(lo))) == [(0, 0), (63, 31)] and W[0, 0] == 283 / 256 and lo[0, 0] == 1 / 256 and lo[63, 31] == 1 / 256
    assert np.array_equal(ho.round_bf16(hi.astype(np.float32)).astype(np.float64), hi)


@pytest.mark.parametrize('head_dim,hw,p0', [(8, 64, 0), (40, 256, 128), (64, 128, 0), (80, 256, 0), (120, 64, 0), (160, 256, 128),
                                            (256, 128, 0), (160, 200, 128)])
def test_tap_chunk_data_path_model(head_dim, hw, p0):
    """The index arithmetic of tap_chunk_kernel (LDS-DMA sources with the swizzle on the source side, clamped rows, partial last
    chunk, Q-operand masks, k-step skip), modelled lane by lane on the host (tools/emulate_tap_chunk.py), reproduces q . k for
    every (pixel, token) of a workgroup tile of the LAST head of the last batch (NaN guard halves behind the tensors: a piece
    past head_dim that was not clamped would show up), and the padded slots stay finite."""
    from tools import emulate_tap_chunk as em
    err, finite = em.check(head_dim, hw, p0=p0, verbose=False)
    assert err < 2e-3 and finite


def test_tap_slab_data_path_model():
    """The index arithmetic of tap_slab_kernel (daam_tap_slab.hip: which instruction of which wave fetches which 16-byte piece into which
    LDS slot -- per-lane offsets shared by instructions ten apart, scalar 16-row steps, the extra instructions of waves 0..3 / 4..7,
    half-size tiles --, the swizzle, the operand reads of every wave role for head_dim 40 / 80 / 160, the zero piece behind the tail
    k-step), modelled lane by lane on the host (tools/emulate_tap_slab.py), reproduces q . k for every (head of the slab, pixel, token) of
    a workgroup of the LAST slab of the last batch (LDS poisoned before the fetches, NaN guard halves behind the tensors)."""
    from tools import emulate_tap_slab as em
    for case in em.CASES:
        err, finite = em.check(verbose=False, **case)
        assert err < 2e-3 and finite, case


def test_committed_counters_are_tied_to_the_build_kernel_by_kernel():
    """profiles/r03_counters.json carries the machine-code fingerprint of every kernel of the build it was measured on; bench.py
    takes its numbers only while each of them is byte-identical in the library built from this tree (source files added since,
    or compiled-out experiments, change ``csrc_sha`` but not the measured kernels) -- and drops them, saying so, as soon as one
    differs (a kernel edit makes the counters stale until the PMC passes are re-run: that is the point, not a failure)."""
    import bench
    from daam_amd import build
    build.build(verbose=False)
    have = build.kernel_shas()
    assert len(have) > 100 and all(len(v) == 12 for v in have.values())
    assert any('tap_d64_kernel' in k for k in have) and any('finalize_up32_pipe_kernel' in k for k in have)
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r03_counters.json')))
    changed = [k for k, v in rec['kernel_shas'].items() if have.get(k) != v]
    prof, note = bench.load_profile('r03_counters.json')
    if changed and rec['csrc_sha'] != build.csrc_sha():
        assert prof is None and 'differ' in note
    else:
        assert prof is not None and 'r03_counters.json' in note
    # the fingerprint is of the code, not of the name: two different kernels never share one
    assert len(set(have.values())) == len(have)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/daam_hip.h compiles as C99 (-pedantic), and a C program -- the shape of the cgo / JNI
    / ctypes stub INTEGRATION.md shows -- links against libdaam_hip.so and calls the entry points that need no GPU."""
    import subprocess
    from daam_amd import _native, build
    build.build(verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'abi.c'
    src.write_text('#include <stdio.h>\n#include "daam_hip.h"\n'
                   'int main(void) {\n'
                   '    DaamCtx* ctx = 0;\n'
                   '    int v = daam_abi_version();\n'
                   '    int rc = daam_ctx_create(4, 99, 64, 0, &ctx);      /* tokens > 80: refused before any device call */\n'
                   '    printf("%d %d %s\\n", v, rc, daam_last_error());\n'
                   '    return (rc != 0 && ctx == 0) ? 0 : 1;\n'
                   '}\n')
    exe = tmp_path / 'abi'
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(root, 'include'), str(src), '-o', str(exe),
                    '-L' + libdir, '-l:' + os.path.basename(_native.LIB_PATH), '-Wl,-rpath,' + libdir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split(None, 2)
    assert int(out[0]) == _native.ABI_VERSION and int(out[1]) == -1 and 'tokens' in out[2]
