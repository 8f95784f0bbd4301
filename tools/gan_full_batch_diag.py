import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dmcnet_amd
from dmcnet_amd import resnet
import tests.test_hip_parity_full as F
from oracle import dmc_oracle as O
T = F.T; HP = F.HP; OPT = F.OPT; DEV = F.DEV
lib = dmcnet_amd._lib.load()
for mode in ("miopen", "own_f32", "own_x3"):
    resnet.OWN_CONV = mode != "miopen"
    lib.dmc_set_option(b"conv_arith", int(mode == "own_x3"))
    o, m = F._pair(51, True, 143)
    b0 = O.synthetic_batch(seed=144, batch=40, num_segments=3, num_class=51)
    b1 = O.synthetic_batch(seed=145, batch=40, num_segments=3, num_class=51)
    md = O.seeded_dropout_masks(146, o.discriminator, 240)
    mg = O.seeded_dropout_masks(147, o.discriminator, 120)
    oopts = O.make_optimizers(o, lr_d_mult=HP["lr_d_mult"], **OPT)
    for opt in oopts:
        O.adjust_learning_rate(opt, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
    step = T.GanTrainStep(m, 3, HP["lr_cls"], HP["lr_adv_g"], HP["lr_adv_d"], HP["lr_mse"], lr_d_mult=HP["lr_d_mult"], **OPT)
    for i, (b, masks) in enumerate(((b0, md), (b1, mg))):
        o.discriminator.forced_masks = masks
        m.discriminator.forced_masks = masks
        ref = O.gan_train_step(o, oopts[0], oopts[1], oopts[2], b, i, 3, HP["lr_cls"], HP["lr_adv_g"], HP["lr_adv_d"], HP["lr_mse"])
        got = step.step(tuple(t.to(DEV) for t in b), i)
        so, sm = o.state_dict(), m.state_dict()
        keys = [k for k in F.WATCH + F.WATCH_D if ("gen_flow_model" in k) == (i == 1)]
        print(mode, "step", i, {k: "%.2e" % F.rel_err(got[k], ref[k]) for k in ("loss", "loss_cls", "loss_adv", "output", "validity")})
        for k in keys:
            if "running_" in k: continue
            d = float((sm[k].float().cpu() - so[k].float()).abs().max())
            print("   %-50s max|dw| %.3e  (%.2f of an Adam step)" % (k, d, d / F._adam_step(k)))
