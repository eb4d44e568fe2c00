"""Batch assembly on the device: the step immediately before the train-step path.

Mirrors what `Data_loaders/audio_loader.py` does on the host with numpy: frame normalisation `(px - 127) / 128`,
left-right flip and random crop (`:185-245`), and the per-clip mel / audio windows with the `(T, D) -> (D, T)`
transpose of `collate_fn` (`:471-475,508,523`).  Image decoding and `cv2.resize` stay on the host (data pipeline,
out of scope): the inputs here are uint8 frames already resized to `image_rescal_size`."""
import torch

from . import _lib
from .ops import _stream


def _need_cuda(t, dtype):
    if not t.is_cuda:
        raise _lib.ViaiLibraryError("viai batch assembly runs on the GPU only (got a %s tensor); no CPU fallback" % t.device)
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def frames_prep(frames_u8, image_size, crop_x=0, crop_y=0, flip=False, nchw=False):
    """frames_u8: (..., S, S, C) uint8, C = 3 (RGB) or 2 (flow_x, flow_y).  Returns float32 (n, size, size, 4) NHWC4
    (channels >= C zero; what ResNet conv1 consumes), or with nchw=True the reference's layout (..., C, size, size).
    audio_loader.py:192-194 draws crop_x / crop_y / flip once per item; `[:, :, crop_x:.., crop_y:..]` crops rows by
    crop_x and columns by crop_y (:238-241)."""
    lib = _lib.load()
    f = _need_cuda(frames_u8, torch.uint8)
    S, C = f.shape[-2], f.shape[-1]
    assert f.shape[-3] == S, "square frames expected (cv2.resize to image_rescal_size)"
    n = f.numel() // (S * S * C)
    out = torch.empty((n, image_size, image_size, 4), dtype=torch.float32, device=f.device)
    _lib.check(lib.viai_frames_prep(f.data_ptr(), out.data_ptr(), n, S, C, image_size, int(crop_x), int(crop_y), int(bool(flip)),
                                    _stream()), "viai_frames_prep")
    if nchw:
        return out[..., :C].permute(0, 3, 1, 2).reshape(tuple(f.shape[:-3]) + (C, image_size, image_size)).contiguous()
    return out


def slice_clips(c, x, starts, use_image_num, hop_size):
    """c: (T_total, D) mel frames, x: (samples,) waveform of one utterance, starts: video-frame start index per clip.
    Returns (c_batch (B, D, 4N), x_batch (B, 1, 4N*hop)): clip b covers mel frames [3 + 4*start, +4N)
    (audio_loader.py:471-475) in the channel-first layout of collate_fn (:508,:523)."""
    lib = _lib.load()
    c = _need_cuda(c, torch.float32)
    x = _need_cuda(x, torch.float32)
    st = torch.as_tensor(starts, dtype=torch.int32, device=c.device).contiguous()
    B, (T_total, D), L = st.numel(), c.shape, 4 * int(use_image_num)
    c_out = torch.empty((B, D, L), dtype=torch.float32, device=c.device)
    x_out = torch.empty((B, 1, L * hop_size), dtype=torch.float32, device=c.device)
    _lib.check(lib.viai_slice_clips(c.data_ptr(), x.data_ptr(), st.data_ptr(), c_out.data_ptr(), x_out.data_ptr(),
                                    B, D, L, int(hop_size), T_total, x.numel(), _stream()), "viai_slice_clips")
    return c_out, x_out
