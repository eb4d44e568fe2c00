"""calibration aid: worst per-tensor gradient errors of the HIP step vs fp64 oracle / fp32 oracle / reference goldens"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import viai_oracle as O
import test_networks_gpu as T
for name, (B, F, Tm) in (("tiny", (2, 80, 32)), ("cfg1", (4, 128, 128))):
    s = O.cf_uniform("s.%s" % name, (B, 1, F, Tm)); mask = O.make_mask(B, Tm, "mask.%s" % name)
    gold = np.load("tests/golden/step_%s.npz" % name)
    m = T.build_model(F, Tm); m.set_inputs(s, mask); m.forward_backward_no_update(); torch.cuda.synchronize()
    wn = ws = 0
    for mod, grp in ((m.netD, "grads_D"), (m.Mel_Encoder, "grads_E"), (m.Mel_Decoder, "grads_G")):
        for k, g in T.named_grads(mod).items():
            gk = "nu.%s.%s.dg" % (grp, k)
            if gk not in gold.files or (grp == "grads_G" and k in T.SHADOWED): continue
            dg, ref = O.digest(g), gold[gk]
            wn = max(wn, abs(dg[2] - ref[2]) / ref[2]); ws = max(ws, np.linalg.norm(dg[3:] - ref[3:]) / (np.linalg.norm(ref[3:]) + 1e-12))
    print(name, "vs reference digests: worst norm err %.2e, worst sample err %.2e" % (wn, ws), flush=True)
    s2 = T.separated_input(s, mask)
    m = T.build_model(F, Tm); m.set_inputs(s2, mask); m.forward_backward_no_update(); torch.cuda.synchronize()
    oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
    ocap = O.step_no_update(oE, oG, oD, s2, mask)
    dcap = O.step_no_update(T.to64(O.encoder_state()), T.to64(O.decoder_state()), T.to64(O.disc_state()), s2.double(), mask.double())
    print(name, "aggregate (e_hip, e_o32) vs fp64 per network:", {k: ("%.2e" % v[0], "%.2e" % v[1]) for k, v in T.check_against_oracle(m, ocap, dcap, oE, oG, oD).items()}, flush=True)
    tie = (m.fake.detach().cpu() - s).abs()
    print(name, "L1 near-ties on the ORIGINAL input: |fake - s| < 1e-6: %d, < 1e-5: %d of %d" % (0, 0, 0) if False else "", flush=True)
    worst = (0, 0, 0, "")
    for mod, grp in ((m.netD, "grads_D"), (m.Mel_Encoder, "grads_E"), (m.Mel_Decoder, "grads_G")):
        for k, g in T.named_grads(mod).items():
            truth = dcap[grp][k]
            if truth is None or (grp == "grads_G" and k in T.SHADOWED): continue
            eh, eo = T.relerr(g, truth), T.relerr(ocap[grp][k], truth)
            margin = eh - 4 * eo
            if margin > worst[0]: worst = (margin, eh, eo, grp + "." + k)
    print(name, "worst per-tensor (e_hip - 4 e_o32) = %.2e  (e_hip %.2e, e_o32 %.2e) at %s" % worst, flush=True)
