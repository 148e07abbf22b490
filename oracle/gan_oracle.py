"""CPU oracle for the Pix2Pix / CycleGAN training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gan_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / reported CPU baseline.

PARITY UNPINNED: the reference (kingjosephm/GAN) has no tests, golden vectors
or fixtures for this path and its arithmetic lives in
tensorflow==2.6.0 / keras==2.6.0 (requirements.txt:116, :53), which is not
installed in this image (``import tensorflow`` -> ModuleNotFoundError).  This
file restates the published TF/Keras 2.6 semantics of the layers the
reference instantiates; it is cross-checked in tests/ against an independent
PyTorch-CPU autograd implementation of the same graph (a second opinion, not
the reference) and against analytic known-answer values.

Every function cites the reference file:line it follows.  All tensors NHWC;
conv kernels HWIO (kh,kw,cin,cout); transposed-conv kernels (kh,kw,cout,cin)
as Keras stores them.  ``dtype`` float32 restates the reference's precision,
float64 is used by the finite-difference and tolerance tests.
"""
from __future__ import annotations

import numpy as np

LEAKY_ALPHA = 0.3   # Keras LeakyReLU() default alpha (base_gan.py:87,155)
BN_EPS = 1e-3       # Keras BatchNormalization default epsilon (base_gan.py:83)
BN_MOMENTUM = 0.99  # Keras BatchNormalization default momentum
IN_EPS = 1e-5       # utils.py:9
ADAM_EPS = 1e-7     # Keras Adam default epsilon (base_gan.py:252)


# --------------------------------------------------------------------------
# convolution primitives
# --------------------------------------------------------------------------
def _im2col(xp, kh, kw, stride, Ho, Wo):
    N, H, W, C = xp.shape
    s = xp.strides
    cols = np.lib.stride_tricks.as_strided(
        xp, (N, Ho, Wo, kh, kw, C),
        (s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]), writeable=False)
    return cols.reshape(N * Ho * Wo, kh * kw * C)


def _col2im(dcols, pshape, kh, kw, stride, Ho, Wo):
    N, H, W, C = pshape
    d = dcols.reshape(N, Ho, Wo, kh, kw, C)
    out = np.zeros(pshape, dtype=dcols.dtype)
    for i in range(kh):
        for j in range(kw):
            out[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride, :] += d[:, :, :, i, j, :]
    return out


def conv2d_fwd(x, w, stride, pad=1):
    """Conv2D k4, zero pad `pad` on every side.  stride 2 + pad 1 == Keras
    padding='same' for even H (base_gan.py:77-79); stride 1 + pad 1 ==
    ZeroPadding2D() followed by a 'valid' conv (base_gan.py:145-148,157-161)."""
    kh, kw, ci, co = w.shape
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    N, H, W, C = xp.shape
    Ho = (H - kh) // stride + 1
    Wo = (W - kw) // stride + 1
    cols = _im2col(xp, kh, kw, stride, Ho, Wo)
    y = cols @ w.reshape(kh * kw * ci, co)
    return y.reshape(N, Ho, Wo, co)


def conv2d_bwd(x, w, dy, stride, pad=1, need_dx=True):
    kh, kw, ci, co = w.shape
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    N, Ho, Wo, _ = dy.shape
    cols = _im2col(xp, kh, kw, stride, Ho, Wo)
    dy2 = dy.reshape(-1, co)
    dw = (cols.T @ dy2).reshape(kh, kw, ci, co)
    dx = None
    if need_dx:
        dcols = dy2 @ w.reshape(kh * kw * ci, co).T
        dxp = _col2im(dcols, xp.shape, kh, kw, stride, Ho, Wo)
        dx = dxp[:, pad:xp.shape[1] - pad, pad:xp.shape[2] - pad, :]
    return dx, dw


def convT2d_fwd(x, w):
    """Conv2DTranspose k4 s2 'same' (base_gan.py:106-110,201-204):
    out[n, 2i+kh-1, 2j+kw-1, co] += x[n,i,j,ci] * w[kh,kw,co,ci]."""
    kh, kw, co, ci = w.shape
    N, h, wd, _ = x.shape
    cols = x.reshape(-1, ci) @ w.transpose(3, 0, 1, 2).reshape(ci, kh * kw * co)
    outp = _col2im(cols, (N, 2 * h + 2, 2 * wd + 2, co), kh, kw, 2, h, wd)
    return outp[:, 1:-1, 1:-1, :]


def convT2d_bwd(x, w, dy, need_dx=True):
    kh, kw, co, ci = w.shape
    N, h, wd, _ = x.shape
    dyp = np.pad(dy, ((0, 0), (1, 1), (1, 1), (0, 0)))
    cols = _im2col(dyp, kh, kw, 2, h, wd)               # (M, kh*kw*co)
    dw = (cols.T @ x.reshape(-1, ci)).reshape(kh, kw, co, ci)
    dx = None
    if need_dx:
        dx = (cols @ w.reshape(kh * kw * co, ci)).reshape(N, h, wd, ci)
    return dx, dw


# --------------------------------------------------------------------------
# normalisation / activation
# --------------------------------------------------------------------------
def _moments(y, kind):
    if kind == 'batchnorm':      # Keras BatchNormalization, training=True: batch stats over N,H,W
        axes = (0, 1, 2)
    else:                        # utils.py:27 tf.nn.moments(x, axes=[1,2], keepdims=True)
        axes = (1, 2)
    mean = y.mean(axis=axes, keepdims=True)
    var = ((y - mean) ** 2).mean(axis=axes, keepdims=True)
    return mean, var


def norm_fwd(y, gamma, beta, kind):
    eps = BN_EPS if kind == 'batchnorm' else IN_EPS
    mean, var = _moments(y, kind)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (y - mean) * rstd
    return gamma * xhat + beta, (xhat, rstd, mean, var)


def norm_bwd(dz, cache, gamma, kind):
    xhat, rstd, _, _ = cache
    axes = (0, 1, 2) if kind == 'batchnorm' else (1, 2)
    dgamma = (dz * xhat).sum(axis=(0, 1, 2))
    dbeta = dz.sum(axis=(0, 1, 2))
    m1 = dz.mean(axis=axes, keepdims=True)
    m2 = (dz * xhat).mean(axis=axes, keepdims=True)
    dy = gamma * rstd * (dz - m1 - xhat * m2)
    return dy, dgamma, dbeta


def act_fwd(z, act):
    if act == 'lrelu':
        return np.where(z > 0, z, LEAKY_ALPHA * z)
    if act == 'relu':
        return np.maximum(z, 0)
    if act == 'tanh':
        return np.tanh(z)
    return z


def act_bwd(da, z, a, act):
    if act == 'lrelu':
        return da * np.where(z > 0, 1.0, LEAKY_ALPHA).astype(da.dtype)
    if act == 'relu':
        return da * (z > 0)
    if act == 'tanh':
        return da * (1 - a * a)
    return da


# --------------------------------------------------------------------------
# losses (base_gan.py:227-245, pix2pix.py:167-188, cycle_gan.py:154-177)
# --------------------------------------------------------------------------
def bce_logits(x, target):
    """BinaryCrossentropy(from_logits=True), mean over all elements
    (base_gan.py:231).  Returns loss and dloss/dx."""
    z = float(target)
    loss = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    sig = np.where(x >= 0, 1 / (1 + np.exp(-np.abs(x))), np.exp(-np.abs(x)) / (1 + np.exp(-np.abs(x))))
    return loss.mean(dtype=np.float64).astype(x.dtype), ((sig - z) / x.size).astype(x.dtype)


def l1_mean(a, b):
    """tf.reduce_mean(tf.abs(a - b)) (pix2pix.py:181); returns loss, d/da."""
    d = a - b
    return np.abs(d).mean(dtype=np.float64).astype(a.dtype), (np.sign(d) / d.size).astype(a.dtype)


# --------------------------------------------------------------------------
# Adam, TF/Keras 2.6 form (base_gan.py:247-252)
# --------------------------------------------------------------------------
class AdamTF:
    def __init__(self, lr=2e-4, beta_1=0.5, beta_2=0.999):
        self.lr, self.b1, self.b2 = lr, beta_1, beta_2
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, params, grads):
        self.t += 1
        f = np.float32 if next(iter(params.values())).dtype == np.float32 else np.float64
        lr_t = f(self.lr) * np.sqrt(f(1) - f(self.b2) ** f(self.t)) / (f(1) - f(self.b1) ** f(self.t))
        for k, g in grads.items():
            p = params[k]
            if k not in self.m:
                self.m[k] = np.zeros_like(p)
                self.v[k] = np.zeros_like(p)
            m, v = self.m[k], self.v[k]
            m += (g - m) * f(1 - self.b1)
            v += (g * g - v) * f(1 - self.b2)
            p -= (m * lr_t) / (np.sqrt(v) + f(ADAM_EPS))


# --------------------------------------------------------------------------
# layer blocks (base_gan.py:63-122)
# --------------------------------------------------------------------------
def block_fwd(x, P, name, kind, stride, norm, act, dropmask=None, training_state=None):
    """conv|convT -> [norm] -> [dropout] -> act.  P: dict of arrays."""
    if kind == 'conv':
        y = conv2d_fwd(x, P[name + '.kernel'], stride)
    else:
        y = convT2d_fwd(x, P[name + '.kernel'])
    if (name + '.bias') in P:
        y = y + P[name + '.bias']
    ncache = None
    z = y
    if norm is not None:
        if norm == 'batchnorm':
            z, ncache = norm_fwd(y, P[name + '.gamma'], P[name + '.beta'], norm)
            if training_state is not None:   # moving-average update, fused-BN Bessel variance
                n = y.shape[0] * y.shape[1] * y.shape[2]
                mean, var = ncache[2].reshape(-1), ncache[3].reshape(-1)
                adj = n / max(n - 1, 1)
                mm = training_state.setdefault(name + '.moving_mean', np.zeros_like(mean))
                mv = training_state.setdefault(name + '.moving_variance', np.ones_like(var))
                mm += (mean - mm) * (1 - BN_MOMENTUM)
                mv += (var * adj - mv) * (1 - BN_MOMENTUM)
        else:
            z, ncache = norm_fwd(y, P[name + '.scale'], P[name + '.offset'], norm)
    zd = z
    if dropmask is not None:                 # Dropout(0.5): survivors x2 (base_gan.py:117-118)
        zd = z * dropmask * 2.0
    a = act_fwd(zd, act)
    return a, dict(x=x, zd=zd, a=a, ncache=ncache, dropmask=dropmask, name=name, kind=kind,
                   stride=stride, norm=norm, act=act)


def block_bwd(da, c, P, grads, need_dx=True):
    name = c['name']
    dzd = act_bwd(da, c['zd'], c['a'], c['act'])
    dz = dzd if c['dropmask'] is None else dzd * c['dropmask'] * 2.0
    if c['norm'] is not None:
        gk, bk = ('.gamma', '.beta') if c['norm'] == 'batchnorm' else ('.scale', '.offset')
        dy, dg, db = norm_bwd(dz, c['ncache'], P[name + gk], c['norm'])
        grads[name + gk] = grads.get(name + gk, 0) + dg
        grads[name + bk] = grads.get(name + bk, 0) + db
    else:
        dy = dz
    if (name + '.bias') in P:
        grads[name + '.bias'] = grads.get(name + '.bias', 0) + dy.sum(axis=(0, 1, 2))
    if c['kind'] == 'conv':
        dx, dw = conv2d_bwd(c['x'], P[name + '.kernel'], dy, c['stride'], need_dx=need_dx)
    else:
        dx, dw = convT2d_bwd(c['x'], P[name + '.kernel'], dy, need_dx=need_dx)
    grads[name + '.kernel'] = grads.get(name + '.kernel', 0) + dw
    return dx


# --------------------------------------------------------------------------
# parameter construction (base_gan.py:74,103,132,200; utils.py:14-24)
# --------------------------------------------------------------------------
G_DOWN = [64, 128, 256, 512, 512, 512, 512, 512]      # base_gan.py:179-188
G_UP = [512, 512, 512, 512, 256, 128, 64]             # base_gan.py:190-198
G_UP_DROPOUT = [True, True, True, False, False, False, False]


def _norm_params(P, name, c, norm, rng, dtype):
    if norm == 'batchnorm':
        P[name + '.gamma'] = np.ones(c, dtype)
        P[name + '.beta'] = np.zeros(c, dtype)
    elif norm == 'instancenorm':
        P[name + '.scale'] = (1.0 + 0.02 * rng.standard_normal(c)).astype(dtype)
        P[name + '.offset'] = np.zeros(c, dtype)


def init_generator(channels, norm='batchnorm', seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    P = {}
    cin = channels
    for i, co in enumerate(G_DOWN):
        P[f'down{i}.kernel'] = (0.02 * rng.standard_normal((4, 4, cin, co))).astype(dtype)
        if i > 0:
            _norm_params(P, f'down{i}', co, norm, rng, dtype)
        cin = co
    for i, co in enumerate(G_UP):
        P[f'up{i}.kernel'] = (0.02 * rng.standard_normal((4, 4, co, cin))).astype(dtype)
        _norm_params(P, f'up{i}', co, norm, rng, dtype)
        cin = co + G_DOWN[6 - i]
    P['last.kernel'] = (0.02 * rng.standard_normal((4, 4, channels, cin))).astype(dtype)
    P['last.bias'] = np.zeros(channels, dtype)
    return P


def init_discriminator(channels, target=True, norm='batchnorm', seed=1, dtype=np.float32):
    rng = np.random.default_rng(seed)
    P = {}
    cin = channels * (2 if target else 1)
    for i, co in enumerate([64, 128, 256]):
        P[f'down{i}.kernel'] = (0.02 * rng.standard_normal((4, 4, cin, co))).astype(dtype)
        if i > 0:
            _norm_params(P, f'down{i}', co, norm, rng, dtype)
        cin = co
    P['conv.kernel'] = (0.02 * rng.standard_normal((4, 4, 256, 512))).astype(dtype)
    _norm_params(P, 'conv', 512, norm, rng, dtype)
    P['last.kernel'] = (0.02 * rng.standard_normal((4, 4, 512, 1))).astype(dtype)
    P['last.bias'] = np.zeros(1, dtype)
    return P


def trainable_count(P):
    return int(sum(v.size for v in P.values()))


# --------------------------------------------------------------------------
# Generator (base_gan.py:168-225)
# --------------------------------------------------------------------------
def generator_fwd(P, x, norm='batchnorm', dropmasks=None, state=None):
    """dropmasks: list of 3 arrays (0/1) shaped like up0..up2 outputs, or None
    for no dropout (rate forced to 0 - used only by deterministic tests)."""
    caches = []
    skips = []
    h = x
    for i in range(8):                                              # base_gan.py:212-214
        h, c = block_fwd(h, P, f'down{i}', 'conv', 2, norm if i > 0 else None, 'lrelu', training_state=state)
        caches.append(c)
        skips.append(h)
    skips = skips[:-1][::-1]                                        # base_gan.py:216
    for i in range(7):                                              # base_gan.py:219-221
        dm = dropmasks[i] if (dropmasks is not None and G_UP_DROPOUT[i]) else None
        h, c = block_fwd(h, P, f'up{i}', 'convT', 2, norm, 'relu', dropmask=dm, training_state=state)
        caches.append(c)
        h = np.concatenate([h, skips[i]], axis=-1)                  # (up, skip) order
    h, c = block_fwd(h, P, 'last', 'convT', 2, None, 'tanh')        # base_gan.py:201-204,223
    caches.append(c)
    return h, caches


def generator_bwd(P, dout, caches, need_dx=False):
    grads = {}
    d = block_bwd(dout, caches[15], P, grads)
    dskips = [None] * 7
    for i in range(6, -1, -1):
        cu = G_UP[i]
        dskips[i] = d[..., cu:]
        d = block_bwd(d[..., :cu], caches[8 + i], P, grads)
    # d is now grad wrt down7 output (bottleneck)
    for i in range(7, -1, -1):
        if i < 7:
            d = d + dskips[6 - i]
        d = block_bwd(d, caches[i], P, grads, need_dx=(i > 0 or need_dx))
    return grads, d


# --------------------------------------------------------------------------
# Discriminator (base_gan.py:124-166)
# --------------------------------------------------------------------------
def discriminator_fwd(P, inp, tar=None, norm='batchnorm', state=None):
    x = inp if tar is None else np.concatenate([inp, tar], axis=-1)   # base_gan.py:137-139
    caches = []
    h = x
    for i in range(3):                                                # base_gan.py:141-143
        h, c = block_fwd(h, P, f'down{i}', 'conv', 2, norm if i > 0 else None, 'lrelu', training_state=state)
        caches.append(c)
    h, c = block_fwd(h, P, 'conv', 'conv', 1, norm, 'lrelu', training_state=state)   # :145-155
    caches.append(c)
    h, c = block_fwd(h, P, 'last', 'conv', 1, None, None)             # :157-161
    caches.append(c)
    return h, caches


def discriminator_bwd(P, dlogits, caches, grads=None, need_dx=False):
    grads = {} if grads is None else grads
    d = dlogits
    for i in range(4, -1, -1):
        d = block_bwd(d, caches[i], P, grads, need_dx=(i > 0 or need_dx))
    return grads, d


# --------------------------------------------------------------------------
# Pix2Pix.train_step (pix2pix.py:190-218)
# --------------------------------------------------------------------------
def pix2pix_train_step(Gp, Dp, optG, optD, inp, tar, lam=100.0, dropmasks=None, training=True,
                       stateG=None, stateD=None, return_grads=False):
    gen, gc = generator_fwd(Gp, inp, 'batchnorm', dropmasks, stateG)            # :200
    d_real, dcr = discriminator_fwd(Dp, inp, tar, 'batchnorm', stateD)          # :202
    d_fake, dcf = discriminator_fwd(Dp, inp, gen, 'batchnorm', stateD)          # :203
    gan_loss, dgan = bce_logits(d_fake, 1.0)                                    # :177
    l1, dl1 = l1_mean(gen, tar)                                                 # :181 |target-gen|
    gen_total = gan_loss + lam * l1                                             # :186
    lr_, dreal = bce_logits(d_real, 1.0)                                        # base_gan.py:241
    lf_, dfake = bce_logits(d_fake, 0.0)                                        # base_gan.py:242
    disc_loss = (lr_ + lf_) * 0.5                                               # :206
    out = (gen_total, gan_loss, l1, disc_loss)
    if not training:
        return out + ((gen,) if return_grads else ())
    # generator gradients: through D(fake) into gen_output (pix2pix.py:210)
    _, dx = discriminator_bwd(Dp, dgan, dcf, grads={}, need_dx=True)
    C = inp.shape[-1]
    dgen = dx[..., C:] + lam * dl1
    gG, _ = generator_bwd(Gp, dgen, gc)
    # discriminator gradients, both branches (pix2pix.py:211)
    gD = {}
    discriminator_bwd(Dp, 0.5 * dreal, dcr, grads=gD)
    discriminator_bwd(Dp, 0.5 * dfake, dcf, grads=gD)
    optG.apply(Gp, gG)                                                          # :213-216
    optD.apply(Dp, gD)
    if return_grads:
        return out + (gen, gG, gD)
    return out


# --------------------------------------------------------------------------
# CycleGAN.train_step (cycle_gan.py:206-276)
# --------------------------------------------------------------------------
def cyclegan_train_step(Gg, Gf, Dx, Dy, opts, real_x, real_y, lam=10.0, dropmasks=None, training=True,
                        return_grads=False):
    """dropmasks: dict with keys 'fake_y','cycled_x','fake_x','cycled_y','same_x','same_y'
    -> list of 3 masks each (one generator call each), or None."""
    dm = (lambda k: None) if dropmasks is None else (lambda k: dropmasks[k])
    n = 'instancenorm'
    fake_y, c_fy = generator_fwd(Gg, real_x, n, dm('fake_y'))       # :220
    cycled_x, c_cx = generator_fwd(Gf, fake_y, n, dm('cycled_x'))   # :221
    fake_x, c_fx = generator_fwd(Gf, real_y, n, dm('fake_x'))       # :223
    cycled_y, c_cy = generator_fwd(Gg, fake_x, n, dm('cycled_y'))   # :224
    same_x, c_sx = generator_fwd(Gf, real_x, n, dm('same_x'))       # :227
    same_y, c_sy = generator_fwd(Gg, real_y, n, dm('same_y'))       # :228
    d_rx, k_rx = discriminator_fwd(Dx, real_x, None, n)             # :230
    d_ry, k_ry = discriminator_fwd(Dy, real_y, None, n)             # :231
    d_fx, k_fx = discriminator_fwd(Dx, fake_x, None, n)             # :233
    d_fy, k_fy = discriminator_fwd(Dy, fake_y, None, n)             # :234
    gen_g_loss, dg_g = bce_logits(d_fy, 1.0)                        # :237
    gen_f_loss, dg_f = bce_logits(d_fx, 1.0)                        # :238
    lcx, dcx = l1_mean(cycled_x, real_x)
    lcy, dcy = l1_mean(cycled_y, real_y)
    total_cycle = lam * lcx + lam * lcy                             # :240
    lsy, dsy = l1_mean(same_y, real_y)
    lsx, dsx = l1_mean(same_x, real_x)
    total_g = gen_g_loss + total_cycle + lam * 0.5 * lsy            # :243
    total_f = gen_f_loss + total_cycle + lam * 0.5 * lsx            # :244
    a, da_rx = bce_logits(d_rx, 1.0)
    b, da_fx = bce_logits(d_fx, 0.0)
    disc_x = (a + b) * 0.5                                          # :246
    a2, da_ry = bce_logits(d_ry, 1.0)
    b2, da_fy = bce_logits(d_fy, 0.0)
    disc_y = (a2 + b2) * 0.5                                        # :247
    out = (gen_g_loss, gen_f_loss, total_cycle, total_g, total_f, disc_x, disc_y)
    if not training:
        return out

    def acc(dst, src):
        for k, v in src.items():
            dst[k] = dst.get(k, 0) + v

    gG, gF = {}, {}
    # ---- total_gen_g_loss wrt G_g (cycle_gan.py:252-253): terms touching G_g:
    #  gen_g_loss via D_y(fake_y); cycle_x via G_f(fake_y) (G_f const); cycle_y via G_g(fake_x); identity_y
    # ---- total_gen_f_loss wrt G_f (:254-255): gen_f_loss via D_x(fake_x); cycle_y via G_g(fake_x)
    #  (G_g const); cycle_x via G_f(fake_y); identity_x.
    # cycled_x = G_f(G_g(real_x)):  d total_cycle/d cycled_x = lam*dcx
    g_f_from_cx, d_fake_y_cyc = generator_bwd(Gf, lam * dcx, c_cx, need_dx=True)   # grads wrt G_f params AND wrt fake_y
    g_g_from_cy, d_fake_x_cyc = generator_bwd(Gg, lam * dcy, c_cy, need_dx=True)
    acc(gF, g_f_from_cx)      # cycle_x term of total_f wrt G_f
    acc(gG, g_g_from_cy)      # cycle_y term of total_g wrt G_g
    # adversarial terms through discriminators (pre-update D weights)
    _, d_fy_in = discriminator_bwd(Dy, dg_g, k_fy, grads={}, need_dx=True)
    _, d_fx_in = discriminator_bwd(Dx, dg_f, k_fx, grads={}, need_dx=True)
    # fake_y = G_g(real_x): upstream = adversarial + cycle_x path (total_g contains total_cycle)
    g1, _ = generator_bwd(Gg, d_fy_in + d_fake_y_cyc, c_fy)
    acc(gG, g1)
    # fake_x = G_f(real_y): upstream = adversarial + cycle_y path
    g2, _ = generator_bwd(Gf, d_fx_in + d_fake_x_cyc, c_fx)
    acc(gF, g2)
    # identity terms
    g3, _ = generator_bwd(Gg, lam * 0.5 * dsy, c_sy)
    acc(gG, g3)
    g4, _ = generator_bwd(Gf, lam * 0.5 * dsx, c_sx)
    acc(gF, g4)
    # discriminators (:257-260)
    gDx, gDy = {}, {}
    discriminator_bwd(Dx, 0.5 * da_rx, k_rx, grads=gDx)
    discriminator_bwd(Dx, 0.5 * da_fx, k_fx, grads=gDx)
    discriminator_bwd(Dy, 0.5 * da_ry, k_ry, grads=gDy)
    discriminator_bwd(Dy, 0.5 * da_fy, k_fy, grads=gDy)
    opts[0].apply(Gg, gG)                                           # :263-273
    opts[1].apply(Gf, gF)
    opts[2].apply(Dx, gDx)
    opts[3].apply(Dy, gDy)
    if return_grads:
        return out + (dict(fake_y=fake_y, fake_x=fake_x), gG, gF, gDx, gDy)
    return out


# --------------------------------------------------------------------------
# input helpers (base_gan.py:56-61; SURVEY 8d synthetic-input definition)
# --------------------------------------------------------------------------
def normalize(img_u8):
    return img_u8.astype(np.float32) / np.float32(127.5) - np.float32(1)


def synthetic_pair(batch, size, channels, seed=123, dtype=np.float32):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (batch, size, size, channels), dtype=np.uint8)
    b = rng.integers(0, 256, (batch, size, size, channels), dtype=np.uint8)
    return normalize(a).astype(dtype), normalize(b).astype(dtype)


def dropout_masks(batch, size, seed=7, dtype=np.float32):
    """Three 0/1 masks shaped like up0..up2 outputs (2x2, 4x4, 8x8 at size 256; x2 at 512)."""
    rng = np.random.default_rng(seed)
    s = size // 256
    return [(rng.random((batch, 2 * s * (1 << i), 2 * s * (1 << i), 512)) >= 0.5).astype(dtype) for i in range(3)]
