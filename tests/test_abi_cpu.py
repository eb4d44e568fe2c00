"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/viai_hip.h declares;
host-side logic that needs no GPU (geometry queries, module/state_dict parity, arenas, loud failure)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from viai_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "viai_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(viai_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_typed(lib):
    from viai_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libviai_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
    for n in _lib.SIGNATURES:
        assert n in names, "%s is bound but not declared in include/viai_hip.h" % n
    assert lib.viai_abi_version() == _lib.ABI_VERSION >= 10


def test_conv_geometry_queries_match_torch(lib):
    from viai_amd._lib import Conv2dDesc
    cases = [  # N IH IW C1 C2 Cout kh kw sh sw ph pw transposed
        (16, 256, 256, 1, 0, 32, 3, 3, 2, 2, 1, 1, 0), (16, 128, 128, 32, 0, 64, 3, 3, 2, 1, 1, 1, 0),
        (16, 2, 16, 256, 0, 256, 3, 3, 1, 1, 0, 1, 1), (16, 256, 256, 1, 0, 64, 1, 4, 1, 2, 0, 1, 0),
        (2, 5, 7, 64, 64, 32, 3, 3, 1, 1, 1, 1, 1),
    ]
    for c in cases:
        d = Conv2dDesc(*(c + (1, 1, -1, -1)))
        oh, ow = C.c_int(), C.c_int()
        lib.viai_conv2d_out_hw(C.byref(d), C.byref(oh), C.byref(ow))
        x = torch.zeros(1, c[3] + c[4], c[1], c[2])
        if c[12]:
            y = torch.nn.functional.conv_transpose2d(x, torch.zeros(c[3] + c[4], 1, c[6], c[7]), stride=1, padding=(c[10], c[11]))
        else:
            y = torch.nn.functional.conv2d(x, torch.zeros(1, c[3] + c[4], c[6], c[7]), stride=(c[8], c[9]), padding=(c[10], c[11]))
        assert (oh.value, ow.value) == (y.shape[2], y.shape[3]), c
        # packed image: fp32 [Cout][taps][Cin], or three bf16 planes (1.5 floats per weight, channels padded to 32)
        n = (c[3] + c[4]) * c[5] * c[6] * c[7]
        pad32 = lambda v: -(-v // 32) * 32
        n_max = 3 * max(pad32(c[5]) * (c[3] + c[4]), pad32(c[3] + c[4]) * c[5]) * c[6] * c[7] // 2 + 1
        assert n <= lib.viai_conv2d_packed_floats(C.byref(d)) <= max(n, n_max)
        nblk, rows = C.c_int(), C.c_int()
        assert lib.viai_conv2d_stat_geom(C.byref(d), C.byref(nblk), C.byref(rows)) == 0
        M = c[0] * oh.value * ow.value
        assert nblk.value == (M + rows.value - 1) // rows.value
        assert lib.viai_conv2d_wgrad_ws_bytes(C.byref(d)) > 0


def test_invalid_descriptors_are_rejected(lib):
    from viai_amd._lib import Conv2dDesc
    bad = Conv2dDesc(1, 8, 8, 32, 0, 32, 3, 3, 2, 2, 1, 1, 1, 1, 1, -1, -1)        # transposed with stride 2: unsupported
    nblk, rows = C.c_int(), C.c_int()
    assert lib.viai_conv2d_stat_geom(C.byref(bad), C.byref(nblk), C.byref(rows)) != 0
    assert lib.viai_conv2d_wgrad_ws_bytes(C.byref(bad)) == 0
    assert lib.viai_conv2d_fwd(C.byref(bad), 0, 0, 0, 0, 0, 0, 0, 0) != 0


def test_module_shells_have_the_reference_state_dict():
    from oracle import viai_oracle as O
    from viai_amd import networks as N
    for mod, sd in ((N.MelEncoder(), O.encoder_state()), (N.MelDecoder(), O.decoder_state()), (N.MelDiscriminator(), O.disc_state())):
        own = mod.state_dict()
        assert list(own.keys()) == list(sd.keys())
        for k in sd:
            assert tuple(own[k].shape) == tuple(sd[k].shape), k
        mod.load_state_dict(sd)


def test_product_path_fails_loudly_without_gpu():
    from viai_amd import _lib, ops
    from viai_amd.model import AudioModel
    with pytest.raises(_lib.ViaiLibraryError):
        ops.bilinear_ac(torch.zeros(1, 2, 2, 4), (4, 4))
    with pytest.raises(_lib.ViaiLibraryError):
        ops.bce_mean(torch.full((4,), 0.5), 1.0)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            AudioModel(device="cpu")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vision-infused-audio-inpainter-viai_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), fn


def test_flat_arena_views_and_checkpoint_layout(tmp_path):
    """FlatArena keeps nn.Parameters usable; FusedAdam.state_dict() has torch.optim.Adam's layout."""
    from viai_amd.model import FlatArena
    lin = torch.nn.Linear(5, 3)
    w0 = lin.weight.detach().clone()
    arena = FlatArena(list(lin.named_parameters()))
    assert torch.equal(lin.weight.detach(), w0)
    assert lin.weight.grad is not None and lin.weight.grad.data_ptr() == arena.grad.data_ptr()
    lin(torch.ones(2, 5)).sum().backward()
    assert float(arena.grad.abs().sum()) > 0            # autograd accumulated INTO the arena
    arena.zero_grad()
    assert float(arena.grad.abs().sum()) == 0
    arena.flat.add_(1.0)
    assert torch.allclose(lin.weight.detach(), w0 + 1.0)  # parameters are views of the arena


def test_synth_matches_oracle_generator():
    from oracle import viai_oracle as O
    from viai_amd import synth
    assert torch.equal(synth.uniform("abc", (5, 7), -1, 2), O.cf_uniform("abc", (5, 7), -1, 2))
    assert torch.equal(synth.time_mask(4, 64, "m"), O.make_mask(4, 64, "m.r0"))


def test_capture_scratch_is_handed_to_the_capturing_model():
    """ops.scratch_snapshot / scratch_take_new (model._capture): buffers created or grown after the snapshot leave the process-wide
    pool and go to the caller; older ones stay."""
    import torch
    from viai_amd import ops
    saved = dict(ops._scratch_pool)
    try:
        ops._scratch_pool.clear()
        old, grown_old = torch.zeros(4), torch.zeros(4)
        ops._scratch_pool[("a", 0, 1)] = old
        ops._scratch_pool[("b", 0, 1)] = grown_old
        before = ops.scratch_snapshot()
        grown_new, fresh = torch.zeros(8), torch.zeros(2)
        ops._scratch_pool[("b", 0, 1)] = grown_new          # re-allocated bigger during the capture
        ops._scratch_pool[("c", 0, 2)] = fresh              # first requested during the capture
        taken = ops.scratch_take_new(before)
        assert set(taken) == {("b", 0, 1), ("c", 0, 2)} and taken[("b", 0, 1)] is grown_new and taken[("c", 0, 2)] is fresh
        assert set(ops._scratch_pool) == {("a", 0, 1)} and ops._scratch_pool[("a", 0, 1)] is old
    finally:
        ops._scratch_pool.clear()
        ops._scratch_pool.update(saved)
