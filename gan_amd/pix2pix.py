"""Pix2Pix with the reference's surface (pix2pix.py:27-461): class `Pix2Pix(GAN)`, `train_step`, `fit`,
`predict`, `parse_opt`, `main` — same flags, defaults, assertions, run-directory layout and metric keys; the
arithmetic of `train_step` runs as one captured hipGraph of hand-written MI355X kernels (gan_amd/steps.py)."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from . import data as D
from . import ddp
from .base_gan import GAN
from .checkpoint import Checkpoint, CheckpointManager, latest_checkpoint
from .steps import Pix2PixStep
from .runner import Run, plot_loss_curves, run_epochs, save_panels
from .utils import pix2pix_losses


def _step_state(ctx, st):
    """Device state a warm-up pass of capture() advances besides weights and Adam moments: the fp16 loss-scale state and the
    per-call dropout draw counters (every new batch size captures a new step: without this each capture would move them)."""
    ts = [ctx.ls] if ctx.ls is not None else []
    for call in vars(st).values():
        md = getattr(call, 'mask_draws', None)
        if isinstance(md, torch.Tensor):
            ts.append(md)
    return [(t, t.clone()) for t in ts]


def _restore_step_state(saved):
    for t, v in saved:
        t.copy_(v)


class Pix2Pix(GAN):
    def __init__(self, config):
        super().__init__(config)
        c = int(self.config['channels'])
        seed = int(self.config.get('seed', 123))
        self.generator = super().Generator(shape=(self.config['img_size'], self.config['img_size'], c), seed=seed)
        self.discriminator = super().Discriminator(target=True, seed=seed + 1)
        mk = lambda: super(Pix2Pix, self).optimizer(learning_rate=self.config['learning_rate'], beta_1=self.config['beta_1'],
                                                   beta_2=self.config['beta_2'])
        self.generator_optimizer = mk().bind(self.generator.net.params)
        self.discriminator_optimizer = mk().bind(self.discriminator.net.params)
        self._steps = {}          # (batch, training) -> (Pix2PixStep, graph replay)
        self.dist = ddp.DistInfo(0, 1, self.config.get('device'))      # main() replaces it in a data-parallel run
        self._rng = np.random.default_rng(seed)
        self.sync = None          # gan_amd.ddp.GradSync for data-parallel training

    def enable_data_parallel(self, info, wire='bf16', exchange='allreduce'):
        """One process per GPU (torchrun): gradients are averaged over the ranks with RCCL all-reduces overlapped with the
        backward pass (gan_amd/steps.py bucketed schedule); BatchNorm statistics stay per replica; the augmentation stream and
        the dropout masks differ per rank.  The reference is single-device (base_gan.py:18-19 only prints the GPU count)."""
        self.dist = info
        if info.world > 1:
            self._rng = np.random.default_rng(int(self.config.get('seed', 123)) + 7919 * info.rank)
            self.sync = ddp.GradSync([self.generator.net.params.grad, self.discriminator.net.params.grad],
                                     compress_bf16=(wire == 'bf16'), lib=self.ctx.lib, exchange=exchange)

    # ---- input pipeline (pix2pix.py:34-165) ------------------------------------------------------
    def split_img(self, image_file: str):
        return D.split_img(super().load(image_file, resize=False), self.config['input_img_orient'])

    def random_crop(self, input_image, real_image, height: int, width: int):
        y, x = self._rng.integers(0, input_image.shape[0] - height + 1), self._rng.integers(0, input_image.shape[1] - width + 1)
        return input_image[y:y + height, x:x + width], real_image[y:y + height, x:x + width]

    def random_jitter(self, input_image, real_image):
        return D.random_jitter_pair(input_image, real_image, self.config['img_size'], self._rng)

    def process_images_train(self, image_file: str):
        a, b = self.split_img(image_file)
        a, b = self.random_jitter(a, b)
        return super().normalize(a), super().normalize(b)

    def process_images_pred(self, image_file: str):
        a, b = self.split_img(image_file)
        s = self.config['img_size']
        return super().normalize(super().resize(a, s, s)), super().normalize(super().resize(b, s, s))

    def image_pipeline(self, predict: bool = False):
        print("\nReading in and processing images.\n", flush=True)
        contents = [i for i in os.listdir(self.config['data']) if 'png' in i or 'jpg' in i]
        assert contents, "No images found in data directory!"
        full = lambda names: [self.config['data'] + '/' + i for i in names]
        dev = self.ctx.device
        if predict:
            return D.Batches(full(contents), self.process_images_pred, 1, None), None, None
        train, val, test = D.pix2pix_split(contents, self.config['seed'], self.config['test_img'], self.config['validation_size'])
        train, val = (ddp.shard_files(f, self.dist.rank, self.dist.world) for f in (train, val))     # (every rank made the same split)
        bs = self.config["batch_size"]
        return (D.Batches(full(train), self.process_images_train, bs, dev),
                D.Batches(full(val), self.process_images_pred, bs, dev),
                D.Batches(full(test), self.process_images_pred, bs, dev))

    # ---- losses / step (pix2pix.py:167-218) ------------------------------------------------------
    def generator_loss(self, disc_generated_output, gen_output, target, input_image):
        gan_loss = self.loss_obj(1.0, disc_generated_output)
        assert self.config['generator_loss'] == 'l1', "only the default l1 secondary loss is on the accelerated path"
        gan_loss2 = (torch.as_tensor(target, device=self.ctx.device).float() - gen_output).abs().mean()
        return gan_loss + (self.config['lambda'] * gan_loss2), gan_loss, gan_loss2

    def _step_for(self, batch, training):
        key = (batch, bool(training))
        if key not in self._steps:
            st = Pix2PixStep(self.ctx, batch, self.config['img_size'], int(self.config['channels']), lam=self.config['lambda'],
                             lr=self.config['learning_rate'], beta_1=self.config['beta_1'], beta_2=self.config['beta_2'],
                             seed=int(self.config.get('seed', 123)), mask_stream=0 if training else 16, nets=(self.generator.net, self.discriminator.net))
            st.sync = self.sync
            saved = self._snapshot()       # capture() runs warm-up passes (they also move BatchNorm's moving statistics): undo them
            extra = _step_state(self.ctx, st)
            replay = st.capture(training=training)
            self._restore(saved)
            _restore_step_state(extra)
            self._steps[key] = (st, replay)
        return self._steps[key]

    def _snapshot(self):
        return [(ps, ps.master.clone(), ps.m.clone(), ps.v.clone(), ps.step.clone(), {k: v.clone() for k, v in ps.state.items()})
                for ps in (self.generator.net.params, self.discriminator.net.params)]

    def _restore(self, saved):
        for ps, w, m, v, step, state in saved:
            ps.master.copy_(w); ps.m.copy_(m); ps.v.copy_(v); ps.step.copy_(step)
            for k, t in state.items():
                ps.state[k].copy_(t)
            ps.prepare()

    def train_step(self, input_image, target, training: bool = True):
        """-> (gen_total_loss, gen_gan_loss, gen_gan_loss2, disc_loss) as 0-d device tensors (no host sync)."""
        x = torch.as_tensor(input_image).to(self.ctx.device, torch.float32)
        y = torch.as_tensor(target).to(self.ctx.device, torch.float32)
        st, replay = self._step_for(x.shape[0], training)
        losses = replay(x.contiguous(), y.contiguous())[:4].clone()
        return losses[0], losses[1], losses[2], losses[3]

    # ---- images / loops (pix2pix.py:220-339) -----------------------------------------------------
    def generate_images(self, model, test_input, tar, path_filename: str):
        """Input | ground truth | `model(test_input, training=True)` (batch statistics and dropout on, pix2pix.py:228)."""
        pred = model(test_input, training=True).cpu().numpy()
        host = lambda t: np.asarray(torch.as_tensor(t).cpu())
        save_panels(path_filename, [('Input Image', host(test_input)[0]), ('Ground Truth', host(tar)[0]), ('Predicted Image', pred[0])],
                    gray=self.config['channels'] == '1')

    def fit(self, train_ds, val_ds, test_ds, output_path: str, checkpoint_manager=None):
        print("\nTraining...\n", flush=True)
        it = iter(test_ds)
        example_input, example_target = next(it)
        it.close()
        samples = os.path.join(output_path, 'test_images')
        if self.dist.is_main:
            os.makedirs(samples, exist_ok=True)
        save = checkpoint_manager.save if checkpoint_manager is not None else (lambda: None)
        sample = lambda epoch: self.generate_images(self.generator, example_input[:1], example_target[:1],
                                                    os.path.join(samples, f"epoch_{epoch}.png"))
        if not self.dist.is_main:           # rank 0 alone writes checkpoints and sample images
            save = sample = (lambda *a: None)
        return run_epochs(self.config['epochs'], list(pix2pix_losses()), lambda: train_ds, lambda: val_ds, self.train_step,
                          save, sample, ('Generator Total Loss', 'Discriminator Loss'),
                          epoch_mean=lambda acc, n: ddp.mean_over_ranks(acc, n, self.dist),
                          after_pass=(self.ctx.assert_no_stack_timeout if self.ctx.use_stacks else None))

    def predict(self, predict_ds, output_path: str):
        plot_path = os.path.join(output_path, 'prediction_images')
        os.makedirs(plot_path, exist_ok=False)
        for k, (inp, tar) in enumerate(predict_ds.unbatch()):
            self.generate_images(self.generator, inp[None], tar[None], os.path.join(plot_path, f"img{k}.png"))


def parse_opt(argv=None):
    """Same flags / defaults / assertions as pix2pix.py:341-377 (+ optional --dtype / --device)."""
    argv = sys.argv[1:] if argv is None else argv
    parser = argparse.ArgumentParser()
    parser.add_argument('--data', type=str, help='path to data', required=True)
    parser.add_argument('--output', type=str, help='path to output results', required=True)
    parser.add_argument('--img-size', type=int, default=256, help='image size h,w')
    parser.add_argument('--batch-size', type=int, default=1, help='batch size per replica')
    parser.add_argument('--buffer-size', type=int, default=99999, help='buffer size')
    parser.add_argument('--channels', type=str, default='1', choices=['1', '3'], help='number of color channels to read in and output')
    parser.add_argument('--logging', type=str, default='true', choices=['true', 'false'], help='turn on/off script logging, e.g. for CLI debugging')
    parser.add_argument('--generator-loss', type=str, default='l1', choices=['l1', 'ssim'], help='combined generator loss function')
    parser.add_argument('--input-img-orient', type=str, default='left', choices=['left', 'right'], help='whether input image is on left (i.e. target right) or vice-versa')
    parser.add_argument('--seed', type=int, default=123, help='seed value for random number generator')
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument('--train', action='store_true', help='train model using data')
    group.add_argument('--predict', action='store_true', help='use pretrained weights to make predictions on data')
    parser.add_argument('--save-weights', type=str, default='true', choices=['true', 'false'], help='save model checkpoints and weights')
    parser.add_argument('--epochs', type=int, default=5, help='number of epochs to train', required='--train' in argv)
    parser.add_argument('--lambda', type=int, default=100, help='lambda value for secondary generator loss (L1)')
    parser.add_argument('--validation-size', type=float, default=0.1, help='validation set size as share of number of training images')
    parser.add_argument('--test-img', type=int, default=5, help='number of test images to sample')
    parser.add_argument('--learning-rate', type=float, default=2e-4, help='learning rate for Adam optimizer for generator and discriminator')
    parser.add_argument('--beta-1', type=float, default=0.5, help='exponential decay rate for 1st moment of Adam optimizer for generator and discriminator')
    parser.add_argument('--beta-2', type=float, default=0.999, help='exponential decay rate for 2st moment of Adam optimizer for generator and discriminator')
    parser.add_argument('--weights', type=str, help='path to pretrained model weights for prediction', required='--predict' in argv)
    parser.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'f16', 'f32'], help='MI355X compute/storage dtype (f32 = exact parity path)')
    parser.add_argument('--device', type=str, default='cuda:0')
    parser.add_argument('--dist-backend', type=str, default='nccl', choices=['nccl', 'gloo'],
                        help='under torchrun (one process per GPU): collective backend; nccl = RCCL over xGMI')
    parser.add_argument('--wire', type=str, default='bf16', choices=['bf16', 'f32'], help='gradient all-reduce wire format')
    parser.add_argument('--exchange', type=str, default='allreduce', choices=['allreduce', 'rs_ag'],
                        help='gradient exchange: one all-reduce per bucket, or fp32 reduce-scatter + all-gather in the wire format')
    args = parser.parse_args(argv)
    if args.generator_loss == 'ssim':
        # pix2pix.py:182-184 computes tf.image.ssim(input_image, target): no gradient reaches the generator and the total becomes
        # a (batch,) vector.  That degenerate term is not built here, and training silently with L1
        # under an 'ssim' label would not be a drop-in: refuse.
        parser.error("--generator-loss ssim is not supported by gan_amd (the reference's SSIM term compares input with target and "
                     "carries no gradient, pix2pix.py:182-184); use the default --generator-loss l1")
    assert (args.img_size == 256) or (args.img_size == 512), "img-size currently only supported for 256 x 256 or 512 x 512 pixels!"
    assert (args.validation_size > 0.0 and args.validation_size <= 0.3), "validation size is a proportion and bounded between 0-0.3!"
    assert (args.test_img >= 1), "test-img is an integer and must be >=1!"
    return args


def main(opt):
    """`python pix2pix.py --train ...` on one GPU, or `torchrun --nproc-per-node N pix2pix.py --train ...` for data-parallel
    training: the process group is joined before the first GPU call, rank r owns cuda:r, the training / validation file lists
    are sharded by rank after the seeded split, and rank 0 alone owns the run directory (logs, checkpoints, figures)."""
    info = ddp.init_from_env(opt.device, opt.dist_backend)
    opt.device = info.device or opt.device
    run = Run(opt.output, log_to_file=opt.logging == 'true' and info.is_main, strict_logs=True, writer=info.is_main)
    try:
        p2p = Pix2Pix(vars(opt))
        if opt.train:
            p2p.enable_data_parallel(info, opt.wire, opt.exchange)
        checkpoint = Checkpoint(generator_optimizer=p2p.generator_optimizer, discriminator_optimizer=p2p.discriminator_optimizer,
                                generator=p2p.generator, discriminator=p2p.discriminator)
        run.write_json('config.json', p2p.config)
        if opt.predict:
            if info.is_main:
                dataset, _, _ = p2p.image_pipeline(predict=True)
                checkpoint.restore(latest_checkpoint(opt.weights))
                p2p.predict(dataset, run.root)
        else:
            train, validation, test = p2p.image_pipeline(predict=False)
            manager = (CheckpointManager(checkpoint, os.path.join(run.root, 'training_checkpoints'), max_to_keep=1)
                       if opt.save_weights == 'true' and info.is_main else None)
            train_metrics, val_metrics = p2p.fit(train, validation, test, run.root, checkpoint_manager=manager)
            ddp.assert_replicas_in_sync([p2p.generator.net.params, p2p.discriminator.net.params], info)
            if info.is_main:
                final = run.dir('final_test_imgs', fresh=True)
                for k, (inp, tar) in enumerate(test.unbatch()):
                    p2p.generate_images(p2p.generator, inp[None], tar[None], os.path.join(final, f"img{k}.png"))
                run.write_json('train_metrics.json', train_metrics)
                run.write_json('val_metrics.json', val_metrics)
                plot_loss_curves(train_metrics, val_metrics, 'Pix2Pix', os.path.join(run.root, 'figs'))
                if info.world > 1:
                    print(f"data-parallel run: {info.world} ranks, replicas in sync.")
        print("Done.")
    except BaseException:
        try:
            run.close()
        finally:
            ddp.shutdown(info, failed=True)  # no barrier on the way out of an exception: the peers are inside other collectives
        raise
    run.close()
    ddp.shutdown(info)


if __name__ == '__main__':
    main(parse_opt())
