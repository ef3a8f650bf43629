"""Secondary measurement: one full GAN training step as exp/cips3d/scripts/train.py:334-491 drives it
(bench.full_gan_step; bench.py reports the C2 case in its own JSON line under "full_step").  Prints ms per phase and
images/s for any geometry.  Not the headline metric."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--num-steps", type=int, default=12)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--freeze", action="store_true", help="GeneratorNerfINR_freeze_NeRF (the r256 stages, ffhq_exp.yaml:192-210)")
    ap.add_argument("--diffaug", action="store_true", help="DiffAugment in D (r256 stages)")
    ap.add_argument("--no-aux", action="store_true", help="train_aux_img False (C4)")
    ap.add_argument("--torch-optim", action="store_true", help="torch clip_grad_norm_ + Adam + python EMA instead of the fused tail")
    a = ap.parse_args()
    from bench import full_gan_step
    print(json.dumps(full_gan_step(torch.device("cuda:0"), a.batch, a.img_size, a.num_steps, steps=a.steps, warmup=a.warmup,
                                   freeze=a.freeze, diffaug=a.diffaug, aux=not a.no_aux, torch_optim=a.torch_optim)))


if __name__ == "__main__":
    main()
