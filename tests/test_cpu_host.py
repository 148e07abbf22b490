"""CPU tests of the host layer: C-ABI library loads and exports every declared symbol (no compute calls),
header <-> binding consistency, CLI surface, input pipeline, checkpoint layout, data-parallel gradient
exchange with gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_header_symbol():
    from gan_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'gan_amd.h')).read()
    declared = set(re.findall(r'\b(gan_[a-zA-Z0-9_]+)\s*\(', hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name)
    assert b'gfx950' in lib.gan_version()


def test_planner_sends_the_256_row_launches_to_the_tap_shared_kernels():
    """Host-side planning only (no GPU): which main loop a convolution launch takes (gan_conv_tap_shared, include/gan_amd.h) at the
    BASELINE shapes of Pix2Pix batch 16 - 2 for the stride-2 shapes and the parity sub-GEMMs, 4 for stride 1, 0 for fp32, for the small
    tiles of the inner layers and when the option is off - and that the planner option round-trips."""
    import ctypes as C
    from gan_amd import _lib as L
    lib = L.load()

    def T(n, h, c):
        return L.GanTensor(16, n, h, h, c, c)

    def shared(op, dt, x, y, w_rows, stride):
        d = L.GanConvDesc(dt, stride, x, y, 16, w_rows, None, 0, 0.3, 0, 16, 1 << 40)
        return lib.gan_conv_tap_shared(C.byref(d), op)

    BF16, F32 = 1, 0
    assert L.get_option('conv.tap_share') == 7
    assert shared(0, BF16, T(16, 128, 64), T(16, 64, 128), 128, 2) == 2       # G.down1 forward: conv s2, 256x128 tiles
    assert shared(3, BF16, T(16, 64, 128), T(16, 32, 512), 512, 2) == 2       # G.up5 dgrad: = stride-2 convolution over dy
    assert shared(2, BF16, T(16, 32, 512), T(16, 64, 128), 128, 2) == 2       # G.up5 forward: transposed conv, parity sub-GEMMs
    assert shared(0, BF16, T(32, 32, 256), T(32, 31, 512), 512, 1) == 4       # D conv4 forward: stride 1, four taps per staged tile
    assert shared(1, BF16, T(32, 31, 512), T(32, 32, 256), 256, 1) == 4       # ... and its dgrad
    assert shared(0, F32, T(16, 128, 64), T(16, 64, 128), 128, 2) == 0        # fp32 stays on the one-tap-per-tile kernel
    assert shared(0, BF16, T(16, 8, 512), T(16, 4, 512), 512, 2) == 0         # inner layer: small tiles
    old = L.set_option('conv.tap_share', 0)
    try:
        assert old == 7 and shared(0, BF16, T(16, 128, 64), T(16, 64, 128), 128, 2) == 0
    finally:
        L.set_option('conv.tap_share', old)
    assert shared(0, BF16, T(16, 128, 64), T(16, 64, 128), 128, 2) == 2


def test_product_path_has_no_cpu_fallback_and_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'gan_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no-oracle', ''), f
    from gan_amd import _lib
    from gan_amd.nets import Ctx
    if not torch.cuda.is_available():
        with pytest.raises(_lib.GanAmdError):
            Ctx('cuda:0', 'bf16')


def test_cli_surface_matches_reference():
    from gan_amd import cycle_gan, pix2pix
    o = pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--train', '--epochs', '3'])
    assert (o.img_size, o.batch_size, o.channels, o.generator_loss, o.input_img_orient, o.seed) == (256, 1, '1', 'l1', 'left', 123)
    assert (getattr(o, 'lambda'), o.validation_size, o.test_img, o.learning_rate, o.beta_1, o.beta_2) == (100, 0.1, 5, 2e-4, 0.5, 0.999)
    with pytest.raises(SystemExit):       # --train and --predict are mutually exclusive, one is required
        pix2pix.parse_opt(['--data', 'd', '--output', 'o'])
    with pytest.raises(SystemExit):       # --epochs required with --train
        pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--train'])
    with pytest.raises(SystemExit):       # --weights required with --predict
        pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--predict'])
    with pytest.raises(SystemExit):       # the flag exists (pix2pix.py:351) but the degenerate SSIM term is not built: refused, not silently L1
        pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--train', '--epochs', '1', '--generator-loss', 'ssim'])
    with pytest.raises(AssertionError):
        pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--train', '--epochs', '1', '--img-size', '128'])
    with pytest.raises(AssertionError):
        pix2pix.parse_opt(['--data', 'd', '--output', 'o', '--train', '--epochs', '1', '--validation-size', '0.5'])
    c = cycle_gan.parse_opt(['--input-images', 'x', '--target-images', 'y', '--output', 'o', '--train', '--epochs', '2'])
    assert getattr(c, 'lambda') == 10 and c.batch_size == 1
    with pytest.raises(SystemExit):       # --target-images required with --train
        cycle_gan.parse_opt(['--input-images', 'x', '--output', 'o', '--train', '--epochs', '2'])
    from gan_amd.utils import cyclegan_losses, pix2pix_losses
    assert list(pix2pix_losses()) == ['Generator Total Loss', 'Generator Loss (Primary)', 'Generator Loss (Secondary)', 'Discriminator Loss']
    assert len(cyclegan_losses()) == 7


def test_input_pipeline(tmp_path):
    from PIL import Image
    from gan_amd import data as D
    rng = np.random.default_rng(0)
    for i in range(12):
        Image.fromarray(rng.integers(0, 256, (40, 100), dtype=np.uint8), 'L').save(tmp_path / f"im{i}.png")
    img = D.load(str(tmp_path / "im0.png"), 1)
    assert img.shape == (40, 100, 1) and img.dtype == np.float32
    a, b = D.split_img(img, 'left')
    assert a.shape == (40, 50, 1) and np.array_equal(a, img[:, :50]) and np.array_equal(b, img[:, 50:])
    a2, b2 = D.split_img(img, 'right')
    assert np.array_equal(a2, b) and np.array_equal(b2, a)
    # nearest neighbour with half-pixel centres: src = floor((dst+.5)*in/out)
    r = D.resize_nearest(np.arange(4, dtype=np.float32).reshape(1, 4, 1), 1, 8)
    assert r[0, :, 0].tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    r = D.resize_nearest(np.arange(5, dtype=np.float32).reshape(1, 5, 1), 1, 2)
    assert r[0, :, 0].tolist() == [1, 3]
    assert np.allclose(D.normalize(np.array([0, 127.5, 255], np.float32)), [-1, 0, 1])
    ja, jb = D.random_jitter_pair(a, b, 32, np.random.default_rng(1))
    assert ja.shape == jb.shape == (32, 32, 1)
    files = D.list_images(str(tmp_path))
    tr, va, te = D.pix2pix_split(files, 123, 5, 0.1)
    assert len(te) == 5 and len(va) == 1 and len(tr) == 6 and not (set(tr) & set(va)) and not (set(te) & set(tr + va))
    assert (tr, va, te) == D.pix2pix_split(files, 123, 5, 0.1)       # seeded: deterministic
    mk = lambda f: tuple(D.normalize(D.resize_nearest(x, 16, 16)) for x in D.split_img(D.load(f, 1)))
    ds = D.Batches([str(tmp_path / f) for f in files], mk, 5)
    batches = list(ds)
    assert [bt[0].shape[0] for bt in batches] == [5, 5, 2] and batches[0][0].shape == (5, 16, 16, 1)   # last partial batch kept


def test_checkpoint_layout_roundtrip(tmp_path):
    from gan_amd.checkpoint import Checkpoint, CheckpointManager, latest_checkpoint, tf_variable_key, GEN_LAYERS

    class Obj:
        def __init__(self, seed):
            r = np.random.default_rng(seed)
            self.sd = {'layer_with_weights-0/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE': r.standard_normal((4, 4, 1, 64)).astype(np.float32),
                       'iter/.ATTRIBUTES/VARIABLE_VALUE': np.array([7], np.int64)}

        def state_dict(self):
            return self.sd

        def load_state_dict(self, sd):
            self.loaded = sd

    assert tf_variable_key('generator', GEN_LAYERS, 'up2.gamma') == 'generator/layer_with_weights-10/layer_with_weights-1/gamma/.ATTRIBUTES/VARIABLE_VALUE'
    g, d = Obj(1), Obj(2)
    mgr = CheckpointManager(Checkpoint(generator=g, discriminator=d), str(tmp_path / 'training_checkpoints'), max_to_keep=1)
    p1 = mgr.save()
    p2 = mgr.save()
    assert os.path.basename(p2) == 'ckpt-2' and not os.path.exists(p1 + '.index')          # keep-last-1
    assert sorted(os.listdir(tmp_path / 'training_checkpoints')) == ['checkpoint', 'ckpt-2.data-00000-of-00001', 'ckpt-2.index']
    assert latest_checkpoint(str(tmp_path / 'training_checkpoints')) == p2
    g2, d2 = Obj(3), Obj(4)
    ck = Checkpoint(generator=g2, discriminator=d2, extra=Obj(5)).restore(p2)               # expect_partial semantics
    k = 'layer_with_weights-0/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'
    assert np.array_equal(g2.loaded[k], g.sd[k]) and np.array_equal(d2.loaded[k], d.sd[k]) and ck.save_counter == 2


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gan_amd.ddp import GradSync, shard_batch
    g = torch.Generator().manual_seed(5)
    full = [torch.randn(1000, generator=g), torch.randn(77, generator=g)]          # "gradients" of the whole batch's shards
    per_rank = [[t * (r + 1) for t in full] for r in range(world)]                 # rank r's local gradients
    mine = [t.clone() for t in per_rank[rank]]
    for compress in (False, True):
        bufs = [t.clone() for t in mine]
        sync = GradSync(bufs, compress_bf16=compress)
        if compress:            # bucketed use: two ranges of buffer 0, buffer 1 whole (what the step graphs do)
            handles = []
            for i, lo, hi in ((0, 0, 640), (0, 640, 1000), (1, 0, 77)):
                sync.pack(i, lo, hi)
                handles.append((sync.start(i, lo, hi), i, lo, hi))
            for h, i, lo, hi in handles:
                sync.wait(h)
                sync.unpack(i, lo, hi)
        else:
            sync()
        expect = [sum(per_rank[r][i] for r in range(world)) / world for i in range(2)]     # mean over ranks
        got = [b * sync.grad_scale for b in bufs]
        tol = 2e-2 if compress else 1e-6
        ok = all(torch.allclose(a, b, rtol=tol, atol=tol) for a, b in zip(got, expect))
        q.put((rank, compress, ok, shard_batch(64, rank, world)))
    # 'rs_ag' exchange (fp32 reduce-scatter + all-gather in the wire format): the same means
    for compress in (False, True):
        bufs = [torch.cat([t, torch.zeros((-t.numel()) % 16)]).clone() for t in mine]          # buckets are multiples of 8 * world
        sync = GradSync(bufs, compress_bf16=compress, exchange='rs_ag')
        hs = [(i, sync.start(i)) for i in range(2)]
        for i, h in hs:
            sync.wait(h)
        got = [(sync.wire[i].float() / world if compress else bufs[i] / world)[:full[i].numel()] for i in range(2)]
        expect = [sum(per_rank[r][i] for r in range(world)) / world for i in range(2)]
        tol = 1e-2 if compress else 1e-6
        q.put((rank, 'rs_ag', all(torch.allclose(a, b, rtol=tol, atol=tol) for a, b in zip(got, expect)), shard_batch(64, rank, world)))
    # helpers of the data-parallel CLI runs (gan_amd/pix2pix.py main under torchrun)
    from gan_amd import ddp
    info = ddp.DistInfo(rank, world, 'cpu')
    files = [f"f{i}" for i in range(11)]
    mine_f = ddp.shard_files(files, rank, world)
    acc, n = torch.tensor([2.0, 4.0]) * (rank + 1), rank + 1          # rank 0: one step, rank 1: two steps
    mean = ddp.mean_over_ranks(acc, n, info)

    class PS:
        master = torch.arange(10.0)
    ddp.assert_replicas_in_sync([PS], info)
    PS.master = torch.arange(10.0) + rank
    diverged = False
    try:
        ddp.assert_replicas_in_sync([PS], info)
    except RuntimeError:
        diverged = True
    q.put((rank, 'helpers', (mine_f, mean.tolist(), diverged), None))
    ddp.shutdown(info)


def _ddp_failing_worker(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gan_amd import ddp
    info = ddp.DistInfo(rank, world, 'cpu')
    try:
        if rank == 1:
            raise ValueError("bad input file on this rank")
        dist.all_reduce(torch.ones(1 << 16))          # the healthy rank sits in a gradient exchange
    except BaseException:
        ddp.shutdown(info, failed=True)               # (what pix2pix.main / cycle_gan.main do on the way out of an exception)
        raise
    ddp.shutdown(info)


def test_failing_rank_leaves_without_a_barrier():
    """A rank that raises must not issue a barrier while its peers sit in another collective (a mismatched collective hangs RCCL
    until the watchdog fires): it exits non-zero at once and the launcher tears the job down."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    port = 29300 + os.getpid() % 500
    procs = [ctx.Process(target=_ddp_failing_worker, args=(r, 2, port)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    procs[1].join(timeout=90)
    assert procs[1].exitcode not in (None, 0) and time.time() - t0 < 90      # left promptly, with the error
    procs[0].join(timeout=90)                          # gloo notices the lost peer; torchrun would have killed it anyway
    if procs[0].exitcode is None:
        procs[0].kill()


def test_gradsync_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(10)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    helpers = {r: v for r, kind, v, _ in res if kind == 'helpers'}
    res = [t for t in res if t[1] != 'helpers']
    assert all(ok for _, _, ok, _ in res), res
    shards = {r: s for r, _, _, s in res}
    assert shards[0] == (0, 32) and shards[1] == (32, 32)
    # file shards: disjoint, equal length (every rank runs the same number of steps), every world-th file
    assert helpers[0][0] == ['f0', 'f2', 'f4', 'f6', 'f8'] and helpers[1][0] == ['f1', 'f3', 'f5', 'f7', 'f9']
    # epoch mean over ALL ranks' steps: (2 + 4, 4 + 8) / (1 + 2)
    assert helpers[0][1] == helpers[1][1] == [2.0, 4.0]
    assert helpers[0][2] and helpers[1][2]                  # diverged replicas are detected on every rank


def test_bench_gpus_without_enough_devices_fails_loudly():
    """`python bench.py --gpus N` with no launcher spawns the ranks itself; with fewer than N GPUs visible it must exit
    non-zero instead of silently benchmarking one GPU."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'GPU(s) visible' in r.stderr, (r.returncode, r.stderr[-300:])
    assert '"metric"' not in r.stdout


def test_header_ctypes_binding_and_integration_snippet_agree():
    """include/gan_amd.h (parsed), gan_amd/_lib.py (the binding the product uses) and the ctypes stub printed in
    INTEGRATION.md declare the same structs, field for field; the INTEGRATION.md snippet is executed (library load and
    symbol binding included - no kernel is launched without a GPU)."""
    import ctypes as C
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import gen_binding
    from gan_amd import _lib as L
    structs = gen_binding.parse_structs()
    assert set(structs) >= {'GanTensor', 'GanConvDesc', 'GanWgradDesc', 'GanPrepEntry', 'GanNormDesc', 'GanNormBwdDesc', 'GanActBwdDesc'}
    for name, fields in structs.items():
        cls = getattr(L, name)
        got = [(f, t) for f, t in cls._fields_]
        want = [(f, getattr(C, t[2:]) if t.startswith('C.') else getattr(L, t)) for f, t in fields]     # (c_int32 is an alias)
        assert got == want, (name, got, want)
        if name.endswith('Desc'):
            assert fields[0] == ('struct_size', 'C.c_uint32') and cls().struct_size == C.sizeof(cls)
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r"```python\nimport ctypes as C\n(.*?)```", text, flags=re.S).group(1)
    gen = code[code.index('# --- generated'):code.index('# --- end of generated part')]
    assert gen.split('\n', 1)[1].strip() == gen_binding.ctypes_source(['GanTensor', 'GanConvDesc']).strip()
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec("import ctypes as C\n" + code, ns)          # loads libgan_amd.so and binds gan_conv2d_fwd
    finally:
        os.chdir(cwd)
    assert C.sizeof(ns['GanConvDesc']) == C.sizeof(L.GanConvDesc) and callable(ns['downsample_conv'])
    # a descriptor with the wrong size is refused before anything is read through it
    d = ns['GanConvDesc'](struct_size=C.sizeof(ns['GanConvDesc']) - 16, dtype=1, stride=2)
    assert ns['lib'].gan_conv2d_fwd(C.byref(d), None) == -1


def test_batches_forward_errors_and_release_their_worker(tmp_path):
    """The prefetch worker is CPU-only, hands its exceptions to the consumer and ends when the iterator is abandoned."""
    import threading
    import time
    from gan_amd import data as D
    files = [f"f{i}" for i in range(20)]

    def ok(f):
        return (np.full((4, 4, 1), float(f[1:]), np.float32),)

    def broken(f):
        if f == 'f5':
            raise ValueError("corrupt image f5")
        return ok(f)

    got = [b[0][:, 0, 0, 0].tolist() for b in D.Batches(files, ok, 3)]
    assert sum(got, []) == [float(i) for i in range(20)] and len(got[-1]) == 2          # last partial batch kept
    with pytest.raises(ValueError, match="corrupt image f5"):
        list(D.Batches(files, broken, 2))
    before = threading.active_count()
    it = iter(D.Batches(files, ok, 1, prefetch=1))
    next(it)
    it.close()                                           # like zip() over unequal sets / next(iter(test_ds))
    for _ in range(50):
        if threading.active_count() <= before:
            break
        time.sleep(0.05)
    assert threading.active_count() <= before


def test_tensorbundle_container_structure_and_roundtrip(tmp_path):
    """gan_amd/tfbundle.py: the TensorBundle files tf.train.Checkpoint writes (pix2pix.py:400-403,419-420), restated without
    TensorFlow.  Checked here: CRC-32C known answers and masking, table magic / footer / block checksums / sorted keys with
    prefix compression across restart points and several data blocks, BundleEntryProto fields, the string-tensor
    encoding of the object graph, and a byte-exact round trip."""
    import struct
    from gan_amd import tfbundle as TB
    assert TB._crc32c(b'123456789') == 0xE3069283 and TB._crc32c(b'') == 0                     # CRC-32C check value
    assert TB._crc32c(b'6789', TB._crc32c(b'12345')) == 0xE3069283                              # chainable
    assert TB._unmask(TB._mask(0x12345678)) == 0x12345678 and TB._mask(0) == 0xa282ead8
    rng = np.random.default_rng(0)
    arrays = {f'generator/layer_with_weights-{i}/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE':
              rng.standard_normal((4, 4, 8, 16)).astype(np.float32) for i in range(40)}
    arrays['generator_optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(12345, np.int64)
    arrays['big/.ATTRIBUTES/VARIABLE_VALUE'] = rng.standard_normal((300, 300)).astype(np.float32)
    names = {k: k.split('/.ATTRIBUTES/')[0] for k in arrays}
    prefix = str(tmp_path / 'ckpt-1')
    TB.write_bundle(prefix, arrays, TB.object_graph(names, []))
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) >= 48          # table footer magic
    items = TB.read_table(prefix + '.index')                                                    # verifies every block checksum
    keys = [k for k, _ in items]
    assert keys[0] == b'' and keys == sorted(keys) and len(keys) == len(arrays) + 2            # header + tensors + object graph
    hdr = dict((n, v) for n, _, v in TB._parse(items[0][1]))
    assert hdr[1] == 1 and dict((n, v) for n, _, v in TB._parse(hdr[3]))[1] == 1                # num_shards 1, version.producer 1
    ent = dict((n, v) for n, _, v in TB._parse(dict(items)[b'big/.ATTRIBUTES/VARIABLE_VALUE']))
    assert ent[1] == 1 and ent[5] == 300 * 300 * 4                                              # DT_FLOAT, size in bytes
    got, graph = TB.read_bundle(prefix)
    assert set(got) == set(arrays) and all(np.array_equal(got[k], arrays[k]) and got[k].dtype == arrays[k].dtype for k in arrays)
    nodes = TB.parse_object_graph(graph)
    n = nodes[0]['children']['generator']
    n = nodes[n]['children']['layer_with_weights-7']
    n = nodes[n]['children']['layer_with_weights-0']
    n = nodes[n]['children']['kernel']
    assert nodes[n]['keys'] == ['generator/layer_with_weights-7/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE']
    # a table with many small entries spans several restart intervals and data blocks
    many = [(f'k{i:06d}'.encode(), (b'v%d' % i) * 50) for i in range(5000)]
    TB.write_table(str(tmp_path / 't.index'), many)
    assert TB.read_table(str(tmp_path / 't.index')) == many
    bad = bytearray(open(tmp_path / 't.index', 'rb').read())
    bad[100] ^= 1
    open(tmp_path / 'bad.index', 'wb').write(bad)
    with pytest.raises(ValueError, match="checksum"):
        TB.read_table(str(tmp_path / 'bad.index'))


def test_tensorbundle_string_tensor_checksum_golden(tmp_path):
    """DT_STRING entries (the object graph): TensorFlow's WriteStringTensor (tensor_bundle.cc) extends the checksum with each
    element length as a little-endian uint32 when it fits 32 bits (uint64 only beyond), stores the masked length crc after the
    varint lengths and keeps the SAME running crc over those 4 bytes and the string bytes.  Golden bytes below were derived
    with an independent bit-by-bit CRC-32C (polynomial 0x82F63B78), not with tfbundle's table-driven one."""
    import struct
    from gan_amd import tfbundle as TB

    def crc_bitwise(data, crc=0):
        crc ^= 0xffffffff
        for b in data:
            crc ^= b
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
        return crc ^ 0xffffffff
    vals = [b'ab', b'c' * 300]
    blob, crc = TB._string_tensor_bytes(vals)
    assert blob[:3] == bytes([0x02, 0xac, 0x02])                                   # varint lengths 2, 300
    assert blob[3:7] == bytes.fromhex('7dadccda')                                  # mask(crc32c(u32(2) ++ u32(300))) little-endian
    assert blob[7:] == b'ab' + b'c' * 300 and len(blob) == 309
    assert crc == 0xef56a1ce and TB._mask(crc) == 0xe620c985                       # entry checksum (BundleEntryProto.crc32c, masked)
    lc = crc_bitwise(struct.pack('<I', 2) + struct.pack('<I', 300))
    assert lc == 0xe1529c24 and TB._string_lengths_crc([2, 300]) == lc
    assert crc_bitwise(struct.pack('<Q', 2) + struct.pack('<Q', 300)) == 0xa5a13801 != lc      # the uint64 convention differs
    assert TB._string_lengths_crc([1 << 32]) == crc_bitwise(struct.pack('<Q', 1 << 32))        # > 32 bits: uint64
    # the reader verifies both checksums of the object-graph entry
    prefix = str(tmp_path / 'ckpt-1')
    names = {'a/.ATTRIBUTES/VARIABLE_VALUE': 'a'}
    TB.write_bundle(prefix, {'a/.ATTRIBUTES/VARIABLE_VALUE': np.zeros(4, np.float32)}, TB.object_graph(names, []))
    _, graph = TB.read_bundle(prefix)
    assert graph and TB.parse_object_graph(graph)[0]['children'] == {'a': 1}
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    for pos, what in ((len(data) - 1, "tensor checksum"), (16 + 1 + 1, "string-length checksum")):     # a string byte; a length-crc byte
        bad = bytearray(data)
        bad[pos] ^= 0x40
        open(prefix + '.data-00000-of-00001', 'wb').write(bad)
        with pytest.raises(ValueError, match=what):
            TB.read_bundle(prefix)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    assert TB.read_bundle(prefix)[1] == graph


def test_checkpoint_keys_follow_the_keras_object_graph(tmp_path):
    """Variable naming of the reference's models under tf.train.Checkpoint (which layers are nested Sequentials, which sit
    directly in the functional model: base_gan.py:124-225), Adam slots under the variable's path, legacy (round-1 JSON)
    checkpoints still readable."""
    import json
    from gan_amd.checkpoint import DISC_LAYERS, GEN_LAYERS, Checkpoint, tf_variable_key
    from gan_amd import tfbundle as TB
    K = lambda o, l, p: tf_variable_key(o, l, p).replace('/.ATTRIBUTES/VARIABLE_VALUE', '')
    assert K('generator', GEN_LAYERS, 'down0.kernel') == 'generator/layer_with_weights-0/layer_with_weights-0/kernel'
    assert K('generator', GEN_LAYERS, 'up6.beta') == 'generator/layer_with_weights-14/layer_with_weights-1/beta'
    assert K('generator', GEN_LAYERS, 'last.bias') == 'generator/layer_with_weights-15/bias'                   # plain Conv2DTranspose
    assert K('discriminator', DISC_LAYERS, 'down2.moving_mean') == 'discriminator/layer_with_weights-2/layer_with_weights-1/moving_mean'
    assert K('discriminator', DISC_LAYERS, 'conv.kernel') == 'discriminator/layer_with_weights-3/kernel'      # plain Conv2D ...
    assert K('discriminator', DISC_LAYERS, 'conv.gamma') == 'discriminator/layer_with_weights-4/gamma'        # ... and its own norm layer
    assert K('discriminator_x', DISC_LAYERS, 'last.kernel') == 'discriminator_x/layer_with_weights-5/kernel'

    class Model:
        layers, obj_name = DISC_LAYERS, 'discriminator'

        def __init__(self, seed):
            r = np.random.default_rng(seed)
            self.P = {'conv.kernel': r.standard_normal((4, 4, 8, 4)).astype(np.float32), 'last.bias': r.standard_normal(1).astype(np.float32)}

        def param_names(self):
            return list(self.P)

        def state_dict(self):
            return {tf_variable_key('discriminator', DISC_LAYERS, k).split('/', 1)[1]: v for k, v in self.P.items()}

        def load_state_dict(self, sd):
            self.loaded = sd

    class Opt:
        def __init__(self, model, seed):
            r = np.random.default_rng(seed)
            self.sd = {'iter/.ATTRIBUTES/VARIABLE_VALUE': np.array(3, np.int64)}
            for slot in ('m', 'v'):
                for k, v in model.P.items():
                    self.sd[f'slot/{slot}/{k}'] = r.standard_normal(v.shape).astype(np.float32)

        def state_dict(self):
            return self.sd

        def load_state_dict(self, sd):
            self.loaded = sd

    m = Model(1)
    o = Opt(m, 2)
    ck = Checkpoint(discriminator=m, discriminator_optimizer=o)
    ck.save_counter = 4
    prefix = ck.write(str(tmp_path / 'ckpt-4'))
    arrays, graph = TB.read_bundle(prefix)
    assert 'discriminator/layer_with_weights-3/kernel/.OPTIMIZER_SLOT/discriminator_optimizer/m/.ATTRIBUTES/VARIABLE_VALUE' in arrays
    assert 'discriminator_optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE' in arrays and int(arrays['save_counter/.ATTRIBUTES/VARIABLE_VALUE']) == 4
    nodes = TB.parse_object_graph(graph)
    opt = nodes[nodes[0]['children']['discriminator_optimizer']]
    assert sorted(s for _, s, _ in opt['slots']) == ['m', 'm', 'v', 'v']
    orig, slot, sv = opt['slots'][0]
    assert nodes[orig]['keys'][0].startswith('discriminator/layer_with_weights-') and '.OPTIMIZER_SLOT' in nodes[sv]['keys'][0]
    m2 = Model(7)
    o2 = Opt(m2, 8)
    ck2 = Checkpoint(discriminator=m2, discriminator_optimizer=o2).restore(prefix)
    assert ck2.save_counter == 4
    assert np.array_equal(m2.loaded['layer_with_weights-3/kernel/.ATTRIBUTES/VARIABLE_VALUE'], m.P['conv.kernel'])
    assert np.array_equal(o2.loaded['slot/v/last.bias'], o.sd['slot/v/last.bias']) and int(o2.loaded['iter/.ATTRIBUTES/VARIABLE_VALUE']) == 3
    # round-1 container (JSON index + raw data) is still readable
    a = m.P['conv.kernel']
    with open(tmp_path / 'old.data-00000-of-00001', 'wb') as f:
        f.write(a.tobytes())
    json.dump({'format': 'gan_amd-bundle-v1', 'tensors': {'discriminator/layer_with_weights-3/kernel/.ATTRIBUTES/VARIABLE_VALUE':
              {'dtype': 'float32', 'shape': list(a.shape), 'offset': 0, 'size': a.nbytes}}}, open(tmp_path / 'old.index', 'w'))
    m3 = Model(9)
    Checkpoint(discriminator=m3).restore(str(tmp_path / 'old'))
    assert np.array_equal(m3.loaded['layer_with_weights-3/kernel/.ATTRIBUTES/VARIABLE_VALUE'], a)


def test_product_input_helpers_match_the_independent_fixture_code(tmp_path):
    """tests/golden/make_golden.py builds the committed example-pair fixture with tests/golden/independent_io.py (PIL decode, split at
    w // 2, TF-2 nearest-neighbour resize in exact integer arithmetic) - code that shares nothing with gan_amd/data.py.  The same
    comparison on images of awkward sizes: the product's load / split_img / resize_nearest (base_gan.py:26-54, pix2pix.py:43-52)
    must return the same pixels, for down- and up-scaling and for both orientations."""
    from PIL import Image
    from gan_amd import data as D
    from tests.golden import independent_io as IO
    rng = np.random.default_rng(5)
    for k, (h, w) in enumerate([(37, 90), (512, 1280), (101, 203), (8, 16)]):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f = str(tmp_path / f"i{k}.png")
        Image.fromarray(a, mode='L').save(f)
        img = D.load(f, 1)
        assert np.array_equal(img[..., 0].astype(np.uint8), IO.decode_gray(f))
        left, right = IO.split_left_right(IO.decode_gray(f))
        pl, pr = D.split_img(img, 'left')
        ql, qr = D.split_img(img, 'right')
        assert np.array_equal(pl[..., 0], left) and np.array_equal(pr[..., 0], right)
        assert np.array_equal(ql[..., 0], pr[..., 0]) and np.array_equal(qr[..., 0], pl[..., 0])
        for oh, ow in [(256, 256), (286, 286), (16, 24), (h, w // 2)]:
            assert np.array_equal(D.resize_nearest(pl, oh, ow)[..., 0], IO.resize_nn(left, oh, ow)), (h, w, oh, ow)
            assert np.array_equal(D.resize_nearest(pr, oh, ow)[..., 0], IO.resize_nn(right, oh, ow)), (h, w, oh, ow)
    # the committed fixture holds exactly two 256 x 256 single-channel pairs
    ex = np.load(os.path.join(ROOT, 'tests', 'golden', 'example_pairs_256.npz'))
    assert ex['input_u8'].shape == ex['target_u8'].shape == (2, 256, 256, 1) and ex['input_u8'].dtype == np.uint8


def test_full_size_golden_fixtures_are_complete():
    """tests/golden/golden_full_*.npz (fp64 oracle at BASELINE.json's shapes, tests/golden/make_golden_full.py): every trainable
    tensor of every network has its checksum, its strided sample and - kernels - its post-Adam sample; finite values only."""
    from oracle import gan_oracle as O
    from tests.golden.make_golden_full import CASES, SAMPLE
    nets = {'pix2pix': {'G': O.init_generator(1), 'D': O.init_discriminator(1, True)},
            'cyclegan': {'Gg': O.init_generator(1, 'instancenorm'), 'Gf': O.init_generator(1, 'instancenorm'),
                         'Dx': O.init_discriminator(1, False, 'instancenorm'), 'Dy': O.init_discriminator(1, False, 'instancenorm')}}
    for name, c in CASES.items():
        g = np.load(os.path.join(ROOT, 'tests', 'golden', f'golden_full_{name}.npz'))
        assert g['losses'].shape == ((4,) if c['model'] == 'pix2pix' else (7,)) and np.isfinite(g['losses']).all()
        n = 0
        for prefix, P in nets[c['model']].items():
            for k, v in P.items():
                s = g[f'gsample/{prefix}.{k}']
                assert g[f'gsum/{prefix}.{k}'].shape == (3,) and 0 < s.size <= SAMPLE and np.isfinite(s).all(), (name, prefix, k)
                stride = max(1, v.size // SAMPLE) | 1
                assert s.size == min(SAMPLE, (v.size + stride - 1) // stride), (name, prefix, k)
                if k.endswith('.kernel'):
                    assert g[f'new/{prefix}.{k}'].shape == s.shape
                n += 1
        assert n == (57 if c['model'] == 'pix2pix' else 114)
        side = (c['S'] + 6) // 7
        key = 'gen_sample' if c['model'] == 'pix2pix' else 'fake_y_sample'
        assert g[key].shape == (c['B'], side, side, 1) and np.abs(g[key]).max() <= 1.0
