"""CycleGAN with the reference's surface (cycle_gan.py:27-502): class `CycleGAN(GAN)`, `train_step`, `fit`,
`predict`, `parse_opt`, `main`; the 6 generator and 4 discriminator invocations of one `train_step`, their
shared backward and four Adam updates run as one captured hipGraph (gan_amd/steps.py::CycleGANStep)."""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from datetime import datetime

import numpy as np
import torch

from . import data as D
from .base_gan import GAN
from .checkpoint import Checkpoint, CheckpointManager, latest_checkpoint
from .steps import CycleGANStep
from .utils import cyclegan_losses, make_fig


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
        self._rng = np.random.default_rng(seed)
        self.sync = None

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
                              seed=int(self.config.get('seed', 123)), nets=tuple(m.net for m in self._models()))
            st.sync = self.sync
            saved = [(ps, ps.master.clone(), ps.m.clone(), ps.v.clone(), ps.step.clone()) for ps in (m.net.params for m in self._models())]
            replay = st.capture(training=training)
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
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        prediction = model(test_input, training=True).cpu().numpy()
        test_input = np.asarray(torch.as_tensor(test_input).cpu())
        plt.figure(figsize=(12, 6))
        display_list = [test_input[0], prediction[0]]
        title = ['Input Image', 'Predicted Image']
        for i in range(2):
            plt.subplot(1, 2, i + 1)
            plt.title(title[i])
            if self.config['channels'] == '1':
                plt.imshow(display_list[i][..., 0] * 0.5 + 0.5, cmap=plt.get_cmap('gray'))
            else:
                plt.imshow(np.clip(display_list[i] * 0.5 + 0.5, 0, 1))
            plt.axis('off')
            plt.tight_layout()
        plt.savefig(path_filename, dpi=200)
        plt.close()

    def fit(self, train_X, train_Y, val_X, val_Y, test, output_path: str, checkpoint_manager=None):
        print("\nTraining...\n", flush=True)
        test = next(iter(test))[0]
        start = time.time()
        train_cost_functions, val_cost_functions = cyclegan_losses(), cyclegan_losses()
        keys = list(train_cost_functions.keys())
        for epoch in range(self.config['epochs']):
            mini_batch_count = 1
            tr, va = [], []
            for (image_x,), (image_y,) in zip(train_X, train_Y):          # tf.data.Dataset.zip: stops at the shorter set
                if image_x.shape[0] != image_y.shape[0]:
                    n = min(image_x.shape[0], image_y.shape[0])
                    image_x, image_y = image_x[:n], image_y[:n]
                tr.append(torch.stack(self.train_step(image_x, image_y)))
                if mini_batch_count % 100 == 0:
                    print('.', end='', flush=True)
                mini_batch_count += 1
            for (image_x,), (image_y,) in zip(val_X, val_Y):
                if image_x.shape[0] != image_y.shape[0]:
                    n = min(image_x.shape[0], image_y.shape[0])
                    image_x, image_y = image_x[:n], image_y[:n]
                va.append(torch.stack(self.train_step(image_x, image_y, training=False)))
            trm = torch.stack(tr).mean(0).cpu().tolist()
            vam = torch.stack(va).mean(0).cpu().tolist() if va else [float('nan')] * 7
            for k, a, b in zip(keys, trm, vam):
                train_cost_functions[k].append(a)
                val_cost_functions[k].append(b)
            test_img_path = output_path + '/test_images'
            os.makedirs(test_img_path, exist_ok=True)
            if ((epoch + 1) % 5 == 0) and ((epoch + 1) != self.config['epochs']):
                if checkpoint_manager is not None:
                    checkpoint_manager.save()
                self.generate_images(self.generator_g, test[:1], path_filename=os.path.join(test_img_path, f"epoch_{epoch + 1}.png"))
            if (epoch + 1) == self.config['epochs']:
                if checkpoint_manager is not None:
                    checkpoint_manager.save()
            print(f'\nCumulative training duration at end of epoch {epoch + 1}: {(time.time() - start) / 60:.2f} min')
            print(f"Train X->Y generator loss: {round(train_cost_functions['Total X->Y Generator Loss'][-1], 2)}, "
                  f"val: {round(val_cost_functions['Total X->Y Generator Loss'][-1], 2)}\n")
        return train_cost_functions, val_cost_functions

    def predict(self, predict_ds, output_path: str):
        plot_path = os.path.join(output_path, 'prediction_images')
        os.makedirs(plot_path)
        for img_counter, i in enumerate(predict_ds.unbatch()):
            self.generate_images(self.generator_g, np.expand_dims(i[0], axis=0), plot_path + "/" + f"img{img_counter}.png")


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
    args = parser.parse_args(argv)
    assert (args.img_size == 256) or (args.img_size == 512), "img-size currently only supported for 256 x 256 or 512 x 512 pixels!"
    assert (args.validation_size > 0.0 and args.validation_size <= 0.3), "validation size is a proportion and bounded between 0-0.3!"
    assert (args.test_img >= 1), "test-img is an integer and must be >=1!"
    return args


def main(opt):
    os.makedirs(opt.output, exist_ok=True)
    full_path = opt.output + '/' + datetime.now().strftime("%Y-%m-%d-%Hh%M")
    os.makedirs(full_path, exist_ok=True)
    log_dir = os.path.join(full_path, 'logs')
    os.makedirs(log_dir, exist_ok=True)
    if opt.logging == 'true':
        sys.stdout = open(os.path.join(log_dir, "Log.txt"), "w")
        sys.stderr = sys.stdout
    cgan = CycleGAN(vars(opt))
    checkpoint = Checkpoint(generator_g=cgan.generator_g, generator_f=cgan.generator_f, discriminator_x=cgan.discriminator_x,
                            discriminator_y=cgan.discriminator_y, generator_g_optimizer=cgan.generator_g_optimizer,
                            generator_f_optimizer=cgan.generator_f_optimizer,
                            discriminator_x_optimizer=cgan.discriminator_x_optimizer,
                            discriminator_y_optimizer=cgan.discriminator_y_optimizer)
    with open(os.path.join(log_dir, 'config.json'), 'w') as f:
        json.dump(cgan.config, f)
    if opt.predict:
        prediction_dataset, _, _, _, _ = cgan.image_pipeline(predict=True)
        checkpoint.restore(latest_checkpoint(opt.weights))
        cgan.predict(prediction_dataset, full_path)
    if opt.train:
        train_X, train_Y, val_X, val_Y, test = cgan.image_pipeline(predict=False)
        if opt.save_weights == 'true':
            manager = CheckpointManager(checkpoint, os.path.join(full_path, 'training_checkpoints'), max_to_keep=3)
        else:
            manager = None
        train_metrics, val_metrics = cgan.fit(train_X, train_Y, val_X, val_Y, test, output_path=full_path, checkpoint_manager=manager)
        final_test_imgs = full_path + '/final_test_imgs'
        os.makedirs(final_test_imgs, exist_ok=False)
        for img_counter, i in enumerate(test.unbatch()):
            cgan.generate_images(cgan.generator_g, np.expand_dims(i[0], axis=0), final_test_imgs + "/" + f"img{img_counter}.png")
        with open(os.path.join(log_dir, 'train_metrics.json'), 'w') as f:
            json.dump(train_metrics, f)
        with open(os.path.join(log_dir, 'val_metrics.json'), 'w') as f:
            json.dump(val_metrics, f)
        import pandas as pd
        for key in train_metrics.keys():
            tr = pd.DataFrame(train_metrics[key]).reset_index()
            va = pd.DataFrame(val_metrics[key]).reset_index()
            tr['index'] = tr['index'] + 1
            tr = tr.set_index('index')
            va['index'] = va['index'] + 1
            va = va.set_index('index')
            make_fig(tr, va, title='CycleGAN ' + key, output_path=os.path.join(full_path, 'figs'))
    print("Done.")


if __name__ == '__main__':
    main(parse_opt())
