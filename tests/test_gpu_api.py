"""GPU test of the drop-in host layer: `Pix2Pix` / `CycleGAN` classes with the reference's surface — train_step,
fit (metrics keys, checkpoint cadence, run layout), checkpoint restore, predict."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_pairs(d, n, rng):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (64, 128), dtype=np.uint8), 'L').save(os.path.join(d, f"p{i}.png"))


def test_pix2pix_cli_train_then_predict(tmp_path):
    from gan_amd import pix2pix
    rng = np.random.default_rng(0)
    data = str(tmp_path / 'data')
    _make_pairs(data, 7, rng)
    out = str(tmp_path / 'out')
    opt = pix2pix.parse_opt(['--data', data, '--output', out, '--train', '--epochs', '1', '--batch-size', '2', '--test-img', '1',
                             '--validation-size', '0.2', '--logging', 'false'])
    pix2pix.main(opt)
    run = os.path.join(out, sorted(os.listdir(out))[0])
    assert sorted(os.listdir(run)) == ['figs', 'final_test_imgs', 'logs', 'test_images', 'training_checkpoints']
    tm = json.load(open(os.path.join(run, 'logs', 'train_metrics.json')))
    assert list(tm) == ['Generator Total Loss', 'Generator Loss (Primary)', 'Generator Loss (Secondary)', 'Discriminator Loss']
    assert all(len(v) == 1 and np.isfinite(v[0]) for v in tm.values())
    assert 5 < tm['Generator Total Loss'][0] < 120          # lambda=100 * L1(~0.5) + BCE(~0.7..)
    ck = os.path.join(run, 'training_checkpoints')
    assert sorted(os.listdir(ck)) == ['checkpoint', 'ckpt-1.data-00000-of-00001', 'ckpt-1.index']
    assert len(os.listdir(os.path.join(run, 'figs'))) == 4 and os.listdir(os.path.join(run, 'final_test_imgs')) == ['img0.png']
    # predict from the saved weights
    out2 = str(tmp_path / 'pred')
    opt2 = pix2pix.parse_opt(['--data', data, '--output', out2, '--predict', '--weights', ck, '--logging', 'false'])
    pix2pix.main(opt2)
    run2 = os.path.join(out2, sorted(os.listdir(out2))[0])
    assert len(os.listdir(os.path.join(run2, 'prediction_images'))) == 7


def test_pix2pix_cli_data_parallel_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 pix2pix.py --train` (north_star: batches shard data-parallel behind the kept CLI): two ranks
    on the one GPU of this box over gloo.  The process group is joined before any GPU call, the file lists are sharded by rank,
    the bucketed gradient exchange runs inside the captured step, rank 0 alone writes the run directory, and main() itself
    checks that both replicas end with bit-identical weights (gan_amd.ddp.assert_replicas_in_sync) - a rank that skipped the
    exchange or drew the wrong shard fails the run."""
    import socket
    import subprocess
    import sys
    rng = np.random.default_rng(1)
    data = str(tmp_path / 'data')
    _make_pairs(data, 11, rng)                     # 1 test + 2 validation + 8 training images -> 4 per rank, 2 steps of batch 2
    out = str(tmp_path / 'out')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'pix2pix.py'), '--data', data, '--output', out, '--train', '--epochs', '1',
           '--batch-size', '2', '--test-img', '1', '--validation-size', '0.2', '--logging', 'false', '--dist-backend', 'gloo']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'data-parallel run: 2 ranks, replicas in sync.' in r.stdout
    runs = os.listdir(out)
    assert len(runs) == 1                                        # rank 1 created nothing
    run = os.path.join(out, runs[0])
    assert sorted(os.listdir(run)) == ['figs', 'final_test_imgs', 'logs', 'test_images', 'training_checkpoints']
    tm = json.load(open(os.path.join(run, 'logs', 'train_metrics.json')))
    assert all(len(v) == 1 and np.isfinite(v[0]) for v in tm.values())
    assert sorted(os.listdir(os.path.join(run, 'training_checkpoints'))) == ['checkpoint', 'ckpt-1.data-00000-of-00001', 'ckpt-1.index']


def test_cyclegan_cli_data_parallel_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 cycle_gan.py --train`: two ranks on the one GPU over gloo through the phased schedule (both
    generators' exchanges after the two chains' second backward, both discriminators' after their parameter passes); rank 0
    alone writes the run directory; main() checks that the four networks of both replicas end bit-identical."""
    import socket
    import subprocess
    import sys
    rng = np.random.default_rng(2)
    dx, dy = str(tmp_path / 'X'), str(tmp_path / 'Y')
    _make_singles(dx, 7, rng)                      # 1 test + 1 validation + 5 training images -> 2 per rank
    _make_singles(dy, 7, rng)
    out = str(tmp_path / 'out')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'cycle_gan.py'), '--input-images', dx, '--target-images', dy, '--output', out,
           '--train', '--epochs', '1', '--batch-size', '1', '--test-img', '1', '--validation-size', '0.2', '--logging', 'false',
           '--dist-backend', 'gloo']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'data-parallel run: 2 ranks, replicas in sync.' in r.stdout
    runs = os.listdir(out)
    assert len(runs) == 1                                        # rank 1 created nothing
    tm = json.load(open(os.path.join(out, runs[0], 'logs', 'train_metrics.json')))
    assert len(tm) == 7 and all(len(v) == 1 and np.isfinite(v[0]) for v in tm.values())


def test_pix2pix_class_surface_and_checkpoint_roundtrip(tmp_path):
    from gan_amd.checkpoint import Checkpoint, CheckpointManager, latest_checkpoint
    from gan_amd.pix2pix import Pix2Pix
    cfg = dict(img_size=256, channels='1', learning_rate=2e-4, beta_1=0.5, beta_2=0.999, seed=3, generator_loss='l1',
               input_img_orient='left', batch_size=2, dtype='f32')
    cfg['lambda'] = 100
    p = Pix2Pix(cfg)
    assert p.generator.count_params() == 54_408_833 and p.discriminator.count_params() == 2_764_545   # SURVEY 8c KATs
    x = torch.rand(2, 256, 256, 1) * 2 - 1
    y = torch.rand(2, 256, 256, 1) * 2 - 1
    w0 = p.generator.net.params.master.clone()
    l_eval = [float(v) for v in p.train_step(x, y, training=False)]
    assert torch.equal(w0, p.generator.net.params.master)                       # training=False: no update
    l1 = [float(v) for v in p.train_step(x, y, True)]
    assert not torch.equal(w0, p.generator.net.params.master)
    assert np.allclose(l_eval[:1], l1[:1], rtol=0.2) and abs(l1[3] - np.log(2)) < 0.3
    # the model callables and the standalone loss helpers agree with the fused step's definitions
    gen = p.generator(x, training=True)
    assert gen.shape == (2, 256, 256, 1) and float(gen.abs().max()) <= 1.0
    d_real = p.discriminator([x, y], training=True)
    assert d_real.shape == (2, 30, 30, 1)
    dl = float(p.discriminator_loss(d_real, d_real, 0.5))
    assert np.isfinite(dl)
    # checkpoint round trip incl. Adam slots
    mgr = CheckpointManager(Checkpoint(generator=p.generator, discriminator=p.discriminator, generator_optimizer=p.generator_optimizer,
                                       discriminator_optimizer=p.discriminator_optimizer), str(tmp_path / 'ck'), max_to_keep=1)
    mgr.save()
    q = Pix2Pix(dict(cfg, seed=99))
    assert not torch.equal(q.generator.net.params.master, p.generator.net.params.master)
    Checkpoint(generator=q.generator, discriminator=q.discriminator, generator_optimizer=q.generator_optimizer,
               discriminator_optimizer=q.discriminator_optimizer).restore(latest_checkpoint(str(tmp_path / 'ck')))
    for a, b in ((p.generator, q.generator), (p.discriminator, q.discriminator)):
        assert torch.equal(a.net.params.master, b.net.params.master) and torch.equal(a.net.params.m, b.net.params.m)
        assert int(a.net.params.step) == int(b.net.params.step) == 1
    l2p = [float(v) for v in p.train_step(x, y, False)]
    l2q = [float(v) for v in q.train_step(x, y, False)]
    assert np.allclose(l2p[2], l2q[2], rtol=0.2)      # same weights -> same L1 up to the (independent) dropout draws


def test_cyclegan_class_one_step():
    from gan_amd.cycle_gan import CycleGAN
    cfg = dict(img_size=256, channels='1', learning_rate=2e-4, beta_1=0.5, beta_2=0.999, seed=3, batch_size=1, dtype='bf16')
    cfg['lambda'] = 10
    c = CycleGAN(cfg)
    x = torch.rand(1, 256, 256, 1) * 2 - 1
    y = torch.rand(1, 256, 256, 1) * 2 - 1
    out = [float(v) for v in c.train_step(x, y)]
    assert len(out) == 7 and all(np.isfinite(out))
    assert abs(out[3] - (out[0] + out[2])) < out[3] and out[2] > 0      # total_g = gen_g + cycle + identity
    assert float(c.generator_g(x).abs().max()) <= 1.0


def test_c_caller_of_the_abi():
    """tests/c_abi_smoke.c: a plain C program (no Python, no torch) runs Conv2D + LeakyReLU through include/gan_amd.h,
    checks it against its own scalar loop and checks that a descriptor with a wrong struct_size is refused."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, 'gan_amd', 'c_abi_smoke')
    if not os.path.exists(exe):
        subprocess.run(['make', '-C', os.path.join(root, 'gan_amd', 'csrc'), 'smoke'], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def _make_singles(d, n, rng):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (80, 96), dtype=np.uint8), 'L').save(os.path.join(d, f"s{i}.png"))


def test_cyclegan_cli_train_then_predict(tmp_path):
    """cycle_gan.py --train / --predict end to end (cycle_gan.py:278-376, 416-497): unequal X / Y sets (the zip stops at the
    shorter one), 5 epochs so that the mid-run checkpoint + sample image cadence is exercised, keep-3 manager, restore."""
    from gan_amd import cycle_gan
    rng = np.random.default_rng(1)
    dx, dy = str(tmp_path / 'X'), str(tmp_path / 'Y')
    _make_singles(dx, 7, rng)
    _make_singles(dy, 5, rng)
    out = str(tmp_path / 'out')
    opt = cycle_gan.parse_opt(['--input-images', dx, '--target-images', dy, '--output', out, '--train', '--epochs', '6', '--batch-size', '2',
                               '--test-img', '1', '--validation-size', '0.2', '--logging', 'true'])
    cycle_gan.main(opt)
    run = os.path.join(out, sorted(os.listdir(out))[0])
    assert sorted(os.listdir(run)) == ['figs', 'final_test_imgs', 'logs', 'test_images', 'training_checkpoints']
    assert sorted(os.listdir(os.path.join(run, 'logs'))) == ['Log.txt', 'config.json', 'train_metrics.json', 'val_metrics.json']
    assert 'Cumulative training duration at end of epoch 6' in open(os.path.join(run, 'logs', 'Log.txt')).read()
    tm = json.load(open(os.path.join(run, 'logs', 'train_metrics.json')))
    vm = json.load(open(os.path.join(run, 'logs', 'val_metrics.json')))
    keys = ['X->Y Generator Loss', 'Y->X Generator Loss', 'Total Cycle Loss', 'Total X->Y Generator Loss',
            'Total Y->X Generator Loss', 'Discriminator X Loss', 'Discriminator Y Loss']
    assert list(tm) == keys and list(vm) == keys
    assert all(len(v) == 6 and np.isfinite(v).all() for v in tm.values()) and all(len(v) == 6 for v in vm.values())
    assert os.listdir(os.path.join(run, 'test_images')) == ['epoch_5.png']              # (epoch+1) % 5 == 0 and not the last
    ck = os.path.join(run, 'training_checkpoints')
    assert sorted(os.listdir(ck)) == ['checkpoint', 'ckpt-1.data-00000-of-00001', 'ckpt-1.index', 'ckpt-2.data-00000-of-00001', 'ckpt-2.index']
    assert len(os.listdir(os.path.join(run, 'figs'))) == 7 and os.listdir(os.path.join(run, 'final_test_imgs')) == ['img0.png']
    out2 = str(tmp_path / 'pred')
    cycle_gan.main(cycle_gan.parse_opt(['--input-images', dx, '--output', out2, '--predict', '--weights', ck, '--logging', 'false']))
    run2 = os.path.join(out2, sorted(os.listdir(out2))[0])
    assert len(os.listdir(os.path.join(run2, 'prediction_images'))) == 7
