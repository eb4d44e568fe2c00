"""per-tensor gradient errors of the HIP step vs the fp64 oracle (and the fp32 oracle's own error), tiny / cfg1 shape, whatever arithmetic
mode the environment selects (VIAI_MATH=fp32, VIAI_F16X2=0):   python tools/grad_table.py [tiny|cfg1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import viai_oracle as O
import test_networks_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B, F, Tm = {"tiny": (2, 80, 32), "cfg1": (4, 128, 128)}[name]
s = O.cf_uniform("s.%s" % name, (B, 1, F, Tm)); mask = O.make_mask(B, Tm, "mask.%s" % name)
s2 = T.separated_input(s, mask)
m = T.build_model(F, Tm); m.set_inputs(s2, mask); m.forward_backward_no_update(); torch.cuda.synchronize()
oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
ocap = O.step_no_update(oE, oG, oD, s2, mask)
dcap = O.step_no_update(T.to64(O.encoder_state()), T.to64(O.decoder_state()), T.to64(O.disc_state()), s2.double(), mask.double())
print("fake", T.relerr(m.fake, dcap["fake"]), "loss_d", m.losses[0].item(), dcap["loss_d"].item(), "loss_g", m.losses[1].item(), dcap["loss_g"].item())
for mod, grp in ((m.netD, "grads_D"), (m.Mel_Encoder, "grads_E"), (m.Mel_Decoder, "grads_G")):
    for k, g in T.named_grads(mod).items():
        truth = dcap[grp][k]
        if truth is None: continue
        print("%-8s %-28s hip %.2e   cpu32 %.2e" % (grp, k, T.relerr(g, truth), T.relerr(ocap[grp][k], truth)))
