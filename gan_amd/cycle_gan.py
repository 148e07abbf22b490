"""CycleGAN with the reference's surface (cycle_gan.py:27-502): class `CycleGAN(GAN)`, `train_step`, `fit`,
`predict`, `parse_opt`, `main`; the 6 generator and 4 discriminator invocations of one `train_step`, their
shared backward and four Adam updates run as one captured hipGraph (gan_amd/steps.py::CycleGANStep)."""
from __future__ import annotations

import argparse
import os
import random
import sys

import numpy as np
import torch

from . import data as D
from . import ddp
from .base_gan import GAN
from .checkpoint import Checkpoint, CheckpointManager, latest_checkpoint
from .steps import CycleGANStep
from .runner import Run, plot_loss_curves, run_epochs, save_panels
from .utils import cyclegan_losses


class CycleGAN(GAN):
    def __init__(self, config):
        super().__init__(config)
        c = int(self.config['channels'])
        seed = int(self.config.get('seed', 123))
        n = 'instancenorm'                                                             # cycle_gan.py:30-33
        self.generator_g = super().Generator(norm_type=n, shape=(None, None, c), seed=seed, name='generator_g')
        self.generator_f = super().Generator(norm_type=n, shape=(None, None, c), seed=seed + 1, name='generator_f')
        self.discriminator_x = super().Discriminator(norm_type=n, target=False, seed=seed + 2, name='discriminator_x')
        self.discriminator_y = super().Discriminator(norm_type=n, target=False, seed=seed + 3, name='discriminator_y')
        mk = lambda m: super(CycleGAN, self).optimizer(learning_rate=self.config['learning_rate'], beta_1=self.config['beta_1'],
                                                      beta_2=self.config['beta_2']).bind(m.net.params)
        self.generator_g_optimizer, self.generator_f_optimizer = mk(self.generator_g), mk(self.generator_f)
        self.discriminator_x_optimizer, self.discriminator_y_optimizer = mk(self.discriminator_x), mk(self.discriminator_y)
        self._steps = {}
        self.dist = ddp.DistInfo(0, 1, self.config.get('device'))
        self._rng = np.random.default_rng(seed)
        self.sync = None

    def enable_data_parallel(self, info, wire='bf16', exchange='allreduce'):
        """As Pix2Pix.enable_data_parallel: one process per GPU, gradients of the four networks averaged over the ranks."""
        self.dist = info
        if info.world > 1:
            self._rng = np.random.default_rng(int(self.config.get('seed', 123)) + 7919 * info.rank)
            self.sync = ddp.GradSync([m.net.params.grad for m in self._models()], compress_bf16=(wire == 'bf16'), lib=self.ctx.lib, exchange=exchange)

    def _models(self):
        return (self.generator_g, self.generator_f, self.discriminator_x, self.discriminator_y)

    # ---- input pipeline (cycle_gan.py:40-152) ----------------------------------------------------
    def random_crop(self, image, height: int, width: int):
        y, x = self._rng.integers(0, image.shape[0] - height + 1), self._rng.integers(0, image.shape[1] - width + 1)
        return image[y:y + height, x:x + width]

    def random_jitter(self, image):
        return D.random_jitter_single(image, self.config['img_size'], self._rng)

    def process_images_train(self, image_file: str):
        return (super().normalize(self.random_jitter(super().load(image_file, resize=True))),)

    def process_images_pred(self, image_file: str):
        s = self.config['img_size']
        return (super().normalize(super().resize(super().load(image_file, resize=True), s, s)),)

    def image_pipeline(self, predict: bool = False):
        print("\nReading in and processing images.\n", flush=True)
        cx = [i for i in os.listdir(self.config['input_images']) if 'png' in i or 'jpg' in i]
        assert cx, "No images found in input image directory!"
        fx = lambda names: [self.config['input_images'] + '/' + i for i in names]
        if predict:
            return D.Batches(fx(cx), self.process_images_pred, 1, None), None, None, None, None
        cy = [i for i in os.listdir(self.config['target_images']) if 'png' in i or 'jpg' in i]
        assert cy, "No images found in target image directory!"
        fy = lambda names: [self.config['target_images'] + '/' + i for i in names]
        random.seed(self.config['seed'])
        test = random.sample(cx, self.config['test_img'])
        val_obs_X = int(np.ceil((len(cx) - self.config['test_img']) * self.config['validation_size']))
        val_obs_Y = int(np.ceil(len(cy) * self.config['validation_size']))
        val_X = random.sample([i for i in cx if i not in test], val_obs_X)
        val_Y = random.sample([i for i in cy], val_obs_Y)
        train_X = [i for i in cx if i not in test and i not in val_X]
        train_Y = [i for i in cy if i not in val_Y]
        train_X, train_Y, val_X, val_Y = (ddp.shard_files(f, self.dist.rank, self.dist.world) for f in (train_X, train_Y, val_X, val_Y))
        bs, dev, sd = self.config["batch_size"], self.ctx.device, self.config['seed']
        return (D.Batches(fx(train_X), self.process_images_train, bs, dev, shuffle_seed=sd),
                D.Batches(fy(train_Y), self.process_images_train, bs, dev, shuffle_seed=sd + 1),
                D.Batches(fx(val_X), self.process_images_pred, bs, dev, shuffle_seed=sd + 2),
                D.Batches(fy(val_Y), self.process_images_pred, bs, dev, shuffle_seed=sd + 3),
                D.Batches(fx(test), self.process_images_pred, bs, dev))

    # ---- losses (cycle_gan.py:154-177) -----------------------------------------------------------
    def generator_loss(self, generated):
        return self.loss_obj(1.0, generated)

    def calc_cycle_loss(self, real_image, cycled_image):
        return (torch.as_tensor(real_image).float() - torch.as_tensor(cycled_image).float()).abs().mean() * self.config['lambda']

    def identity_loss(self, real_image, same_image):
        return self.config['lambda'] * 0.5 * (torch.as_tensor(real_image).float() - torch.as_tensor(same_image).float()).abs().mean()

    # ---- step (cycle_gan.py:206-276) -------------------------------------------------------------
    def _step_for(self, batch, training):
        key = (batch, bool(training))
        if key not in self._steps:
            st = CycleGANStep(self.ctx, batch, self.config['img_size'], int(self.config['channels']), lam=self.config['lambda'],
                              lr=self.config['learning_rate'], beta_1=self.config['beta_1'], beta_2=self.config['beta_2'],
                              seed=int(self.config.get('seed', 123)), mask_stream=0 if training else 16, nets=tuple(m.net for m in self._models()))
            st.sync = self.sync
            saved = [(ps, ps.master.clone(), ps.m.clone(), ps.v.clone(), ps.step.clone()) for ps in (m.net.params for m in self._models())]
            from .pix2pix import _step_state, _restore_step_state
            extra = _step_state(self.ctx, st)
            replay = st.capture(training=training)
            _restore_step_state(extra)
            for ps, w, m, v, step in saved:          # capture() runs warm-up steps: undo them
                ps.master.copy_(w); ps.m.copy_(m); ps.v.copy_(v); ps.step.copy_(step); ps.prepare()
            self._steps[key] = (st, replay)
        return self._steps[key]

    def train_step(self, real_x, real_y, training: bool = True):
        """-> the reference's 7 losses (cycle_gan.py:275-276) as 0-d device tensors."""
        x = torch.as_tensor(real_x).to(self.ctx.device, torch.float32).contiguous()
        y = torch.as_tensor(real_y).to(self.ctx.device, torch.float32).contiguous()
        st, replay = self._step_for(x.shape[0], training)
        return tuple(replay(x, y)[:7].clone().unbind(0))

    # ---- images / loops (cycle_gan.py:179-204, 278-376) ------------------------------------------
    def generate_images(self, model, test_input, path_filename: str):
        """Input | `model(test_input, training=True)` (cycle_gan.py:186)."""
        pred = model(test_input, training=True).cpu().numpy()
        save_panels(path_filename, [('Input Image', np.asarray(torch.as_tensor(test_input).cpu())[0]), ('Predicted Image', pred[0])],
                    gray=self.config['channels'] == '1')

    @staticmethod
    def _zipped(ds_x, ds_y):
        """tf.data.Dataset.zip of the two unpaired sets (cycle_gan.py:297): stops at the shorter one; a last partial batch
        on one side is cut to the other's size.  Both iterators are closed, so their decode threads end with the pass."""
        ix, iy = iter(ds_x), iter(ds_y)
        try:
            for (x,), (y,) in zip(ix, iy):
                n = min(x.shape[0], y.shape[0])
                yield x[:n], y[:n]
        finally:
            ix.close(); iy.close()

    def fit(self, train_X, train_Y, val_X, val_Y, test, output_path: str, checkpoint_manager=None):
        print("\nTraining...\n", flush=True)
        it = iter(test)
        example = next(it)[0]
        it.close()
        samples = os.path.join(output_path, 'test_images')
        if self.dist.is_main:
            os.makedirs(samples, exist_ok=True)
        save = checkpoint_manager.save if checkpoint_manager is not None else (lambda: None)
        sample = lambda epoch: self.generate_images(self.generator_g, example[:1], os.path.join(samples, f"epoch_{epoch}.png"))
        if not self.dist.is_main:
            save = sample = (lambda *a: None)
        return run_epochs(self.config['epochs'], list(cyclegan_losses()), lambda: self._zipped(train_X, train_Y),
                          lambda: self._zipped(val_X, val_Y), self.train_step, save, sample,
                          ('Total X->Y Generator Loss', 'Discriminator Y Loss'),
                          epoch_mean=lambda acc, n: ddp.mean_over_ranks(acc, n, self.dist),
                          after_pass=(self.ctx.assert_no_stack_timeout if self.ctx.use_stacks else None))

    def predict(self, predict_ds, output_path: str):
        plot_path = os.path.join(output_path, 'prediction_images')
        os.makedirs(plot_path)
        for k, (img,) in enumerate(predict_ds.unbatch()):
            self.generate_images(self.generator_g, img[None], os.path.join(plot_path, f"img{k}.png"))


def parse_opt(argv=None):
    """Same flags / defaults / assertions as cycle_gan.py:379-414 (+ optional --dtype / --device)."""
    argv = sys.argv[1:] if argv is None else argv
    parser = argparse.ArgumentParser()
    parser.add_argument('--input-images', type=str, help='path to input images', required=True)
    parser.add_argument('--output', type=str, help='path to output results', required=True)
    parser.add_argument('--img-size', type=int, default=256, help='image size h,w')
    parser.add_argument('--batch-size', type=int, default=1, help='batch size')
    parser.add_argument('--buffer-size', type=int, default=99999, help='buffer size')
    parser.add_argument('--channels', type=str, default='1', choices=['1', '3'], help='number of color channels to read in and output')
    parser.add_argument('--logging', type=str, default='true', choices=['true', 'false'], help='turn on/off script logging, e.g. for CLI debugging')
    parser.add_argument('--seed', type=int, default=123, help='seed value for random number generator')
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument('--train', action='store_true', help='train model using data')
    group.add_argument('--predict', action='store_true', help='use pretrained weights to make predictions on data')
    parser.add_argument('--target-images', type=str, help='path to target images', required='--train' in argv)
    parser.add_argument('--epochs', type=int, default=5, help='number of epochs to train', required='--train' in argv)
    parser.add_argument('--validation-size', type=float, default=0.1, help='validation set size as share of number of training images')
    parser.add_argument('--test-img', type=int, default=5, help='number of test images to sample')
    parser.add_argument('--save-weights', type=str, default='true', choices=['true', 'false'], help='save model checkpoints and weights')
    parser.add_argument('--lambda', type=int, default=10, help='lambda parameter value')
    parser.add_argument('--learning-rate', type=float, default=2e-4, help='learning rate for Adam optimizer for generators and discriminators')
    parser.add_argument('--beta-1', type=float, default=0.5, help='exponential decay rate for 1st moment of Adam optimizer for generators and discriminators')
    parser.add_argument('--beta-2', type=float, default=0.999, help='exponential decay rate for 2st moment of Adam optimizer for generators and discriminators')
    parser.add_argument('--weights', type=str, help='path to pretrained model weights for prediction', required='--predict' in argv)
    parser.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'f16', 'f32'])
    parser.add_argument('--device', type=str, default='cuda:0')
    parser.add_argument('--dist-backend', type=str, default='nccl', choices=['nccl', 'gloo'],
                        help='under torchrun (one process per GPU): collective backend; nccl = RCCL over xGMI')
    parser.add_argument('--wire', type=str, default='bf16', choices=['bf16', 'f32'], help='gradient all-reduce wire format')
    parser.add_argument('--exchange', type=str, default='allreduce', choices=['allreduce', 'rs_ag'],
                        help='gradient exchange: one all-reduce per bucket, or fp32 reduce-scatter + all-gather in the wire format')
    args = parser.parse_args(argv)
    assert (args.img_size == 256) or (args.img_size == 512), "img-size currently only supported for 256 x 256 or 512 x 512 pixels!"
    assert (args.validation_size > 0.0 and args.validation_size <= 0.3), "validation size is a proportion and bounded between 0-0.3!"
    assert (args.test_img >= 1), "test-img is an integer and must be >=1!"
    return args


def main(opt):
    """One GPU, or `torchrun --nproc-per-node N cycle_gan.py --train ...` (data parallel, as gan_amd.pix2pix.main)."""
    info = ddp.init_from_env(opt.device, opt.dist_backend)
    opt.device = info.device or opt.device
    run = Run(opt.output, log_to_file=opt.logging == 'true' and info.is_main, strict_logs=False, writer=info.is_main)
    try:
        cgan = CycleGAN(vars(opt))
        if opt.train:
            cgan.enable_data_parallel(info, opt.wire, opt.exchange)
        names = ('generator_g', 'generator_f', 'discriminator_x', 'discriminator_y')
        objects = {n: getattr(cgan, n) for n in names}
        objects.update({n + '_optimizer': getattr(cgan, n + '_optimizer') for n in names})
        checkpoint = Checkpoint(**objects)                     # object names of cycle_gan.py:437-444
        run.write_json('config.json', cgan.config)
        if opt.predict:
            if info.is_main:
                dataset = cgan.image_pipeline(predict=True)[0]
                checkpoint.restore(latest_checkpoint(opt.weights))
                cgan.predict(dataset, run.root)
        else:
            train_X, train_Y, val_X, val_Y, test = cgan.image_pipeline(predict=False)
            manager = (CheckpointManager(checkpoint, os.path.join(run.root, 'training_checkpoints'), max_to_keep=3)
                       if opt.save_weights == 'true' and info.is_main else None)
            train_metrics, val_metrics = cgan.fit(train_X, train_Y, val_X, val_Y, test, run.root, checkpoint_manager=manager)
            ddp.assert_replicas_in_sync([m.net.params for m in cgan._models()], info)
            if info.is_main:
                final = run.dir('final_test_imgs', fresh=True)
                for k, (img,) in enumerate(test.unbatch()):
                    cgan.generate_images(cgan.generator_g, img[None], os.path.join(final, f"img{k}.png"))
                run.write_json('train_metrics.json', train_metrics)
                run.write_json('val_metrics.json', val_metrics)
                plot_loss_curves(train_metrics, val_metrics, 'CycleGAN', os.path.join(run.root, 'figs'))
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
