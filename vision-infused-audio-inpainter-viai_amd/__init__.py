"""viai_amd — MI355X-native hot path of the Vision-Infused Audio Inpainter (VIAI).

Import name: `viai_amd` (the on-disk directory keeps the project's hyphenated
name; the top-level `viai_amd/` shim maps the import name onto it).

  _lib      ctypes binding of libviai_hip.so (C ABI: include/viai_hip.h)
  ops       torch.autograd.Function wrappers (forward + backward in HIP)
  networks  MelEncoder / MelDecoder / MelDiscriminator shells (reference API + state_dict)
  model     AudioModel: the G+D train step (reference: the missing Models/Whole_Sync_inpainting_modify)
  ddp       one-process-per-GPU gradient all-reduce over RCCL
  synth     closed-form synthetic MUSICES-shaped inputs
"""
__version__ = "0.1.0"
