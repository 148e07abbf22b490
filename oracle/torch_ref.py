"""Independent PyTorch-CPU autograd implementation of the same graph as
oracle/gan_oracle.py (a second opinion, NOT the reference).
TEST INFRASTRUCTURE: used by tests/ to cross-check the oracle's explicit backward passes and by bench.py's
cpu_baseline leg as the "PyTorch-CPU eager" stand-in SURVEY.md 8(d) names; never imported by gan_amd/."""
import numpy as np
import torch
import torch.nn.functional as F

LEAKY = 0.3


def t(x, dt=torch.float64, grad=False):
    v = torch.tensor(np.asarray(x), dtype=dt)
    if grad:
        v.requires_grad_(True)
    return v


def nchw(x):
    return x.permute(0, 3, 1, 2)


def nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv(x, w_hwio, stride):
    return nhwc(F.conv2d(nchw(x), w_hwio.permute(3, 2, 0, 1).contiguous(), stride=stride, padding=1))


def convT(x, w_hwoi):
    # torch weight (cin, cout, kh, kw) = w_tf[kh,kw,co,ci] -> permute(3,2,0,1)
    return nhwc(F.conv_transpose2d(nchw(x), w_hwoi.permute(3, 2, 0, 1).contiguous(), stride=2, padding=1))


def norm(y, g, b, kind):
    if kind == 'batchnorm':
        mean = y.mean(dim=(0, 1, 2), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
        eps = 1e-3
    else:
        mean = y.mean(dim=(1, 2), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(1, 2), keepdim=True)
        eps = 1e-5
    return g * (y - mean) * torch.rsqrt(var + eps) + b


def block(x, P, name, kind, stride, nk, act, mask=None):
    y = conv(x, P[name + '.kernel'], stride) if kind == 'conv' else convT(x, P[name + '.kernel'])
    if name + '.bias' in P:
        y = y + P[name + '.bias']
    if nk == 'batchnorm':
        y = norm(y, P[name + '.gamma'], P[name + '.beta'], nk)
    elif nk == 'instancenorm':
        y = norm(y, P[name + '.scale'], P[name + '.offset'], nk)
    if mask is not None:
        y = y * mask * 2.0
    if act == 'lrelu':
        y = F.leaky_relu(y, LEAKY)
    elif act == 'relu':
        y = F.relu(y)
    elif act == 'tanh':
        y = torch.tanh(y)
    return y


def generator(P, x, nk, masks=None):
    skips = []
    h = x
    for i in range(8):
        h = block(h, P, f'down{i}', 'conv', 2, nk if i > 0 else None, 'lrelu')
        skips.append(h)
    skips = skips[:-1][::-1]
    for i in range(7):
        m = masks[i] if (masks is not None and i < 3) else None
        h = block(h, P, f'up{i}', 'convT', 2, nk, 'relu', m)
        h = torch.cat([h, skips[i]], dim=-1)
    return block(h, P, 'last', 'convT', 2, None, 'tanh')


def discriminator(P, inp, tar, nk):
    h = inp if tar is None else torch.cat([inp, tar], dim=-1)
    for i in range(3):
        h = block(h, P, f'down{i}', 'conv', 2, nk if i > 0 else None, 'lrelu')
    h = block(h, P, 'conv', 'conv', 1, nk, 'lrelu')
    return block(h, P, 'last', 'conv', 1, None, None)


def bce(x, target):
    return F.binary_cross_entropy_with_logits(x, torch.full_like(x, target))


def params(Pnp, dt=torch.float64):
    return {k: t(v, dt, grad=True) for k, v in Pnp.items()}


def pix2pix_losses_and_grads(Gnp, Dnp, inp, tar, lam, masks, dt=torch.float64):
    G, D = params(Gnp, dt), params(Dnp, dt)
    x, y = t(inp, dt), t(tar, dt)
    ms = None if masks is None else [t(m, dt) for m in masks]
    gen = generator(G, x, 'batchnorm', ms)
    d_real = discriminator(D, x, y, 'batchnorm')
    d_fake = discriminator(D, x, gen, 'batchnorm')
    gan = bce(d_fake, 1.0)
    l1 = (y - gen).abs().mean()
    gen_total = gan + lam * l1
    disc = (bce(d_real, 1.0) + bce(d_fake, 0.0)) * 0.5
    gG = torch.autograd.grad(gen_total, list(G.values()), retain_graph=True)
    gD = torch.autograd.grad(disc, list(D.values()))
    return ((gen_total.item(), gan.item(), l1.item(), disc.item()), gen.detach().numpy(),
            {k: g.numpy() for k, g in zip(G.keys(), gG)}, {k: g.numpy() for k, g in zip(D.keys(), gD)})


def cyclegan_losses_and_grads(Ggn, Gfn, Dxn, Dyn, rx, ry, lam, masks, dt=torch.float64):
    Gg, Gf, Dx, Dy = params(Ggn, dt), params(Gfn, dt), params(Dxn, dt), params(Dyn, dt)
    x, y = t(rx, dt), t(ry, dt)
    mk = (lambda k: None) if masks is None else (lambda k: [t(m, dt) for m in masks[k]])
    n = 'instancenorm'
    fake_y = generator(Gg, x, n, mk('fake_y'))
    cycled_x = generator(Gf, fake_y, n, mk('cycled_x'))
    fake_x = generator(Gf, y, n, mk('fake_x'))
    cycled_y = generator(Gg, fake_x, n, mk('cycled_y'))
    same_x = generator(Gf, x, n, mk('same_x'))
    same_y = generator(Gg, y, n, mk('same_y'))
    d_rx = discriminator(Dx, x, None, n)
    d_ry = discriminator(Dy, y, None, n)
    d_fx = discriminator(Dx, fake_x, None, n)
    d_fy = discriminator(Dy, fake_y, None, n)
    gen_g = bce(d_fy, 1.0)
    gen_f = bce(d_fx, 1.0)
    cyc = lam * (x - cycled_x).abs().mean() + lam * (y - cycled_y).abs().mean()
    tot_g = gen_g + cyc + lam * 0.5 * (y - same_y).abs().mean()
    tot_f = gen_f + cyc + lam * 0.5 * (x - same_x).abs().mean()
    dxl = (bce(d_rx, 1.0) + bce(d_fx, 0.0)) * 0.5
    dyl = (bce(d_ry, 1.0) + bce(d_fy, 0.0)) * 0.5
    g1 = torch.autograd.grad(tot_g, list(Gg.values()), retain_graph=True)
    g2 = torch.autograd.grad(tot_f, list(Gf.values()), retain_graph=True)
    g3 = torch.autograd.grad(dxl, list(Dx.values()), retain_graph=True)
    g4 = torch.autograd.grad(dyl, list(Dy.values()))
    losses = tuple(v.item() for v in (gen_g, gen_f, cyc, tot_g, tot_f, dxl, dyl))
    pk = lambda P, g: {k: v.numpy() for k, v in zip(P.keys(), g)}
    return losses, pk(Gg, g1), pk(Gf, g2), pk(Dx, g3), pk(Dy, g4)


def pix2pix_train_step_eager(G, D, optstate, inp, tar, lam, masks, lr=2e-4, b1=0.5, b2=0.999, eps=1e-7):
    """One eager PyTorch-CPU Pix2Pix train_step (pix2pix.py:190-218) on torch parameter dicts, TF-form Adam included
    (timing stand-in for the TF CPU path; fp32)."""
    gen = generator(G, inp, 'batchnorm', masks)
    d_real = discriminator(D, inp, tar, 'batchnorm')
    d_fake = discriminator(D, inp, gen, 'batchnorm')
    gan = bce(d_fake, 1.0)
    l1 = (tar - gen).abs().mean()
    gen_total = gan + lam * l1
    disc = (bce(d_real, 1.0) + bce(d_fake, 0.0)) * 0.5
    gG = torch.autograd.grad(gen_total, list(G.values()), retain_graph=True)
    gD = torch.autograd.grad(disc, list(D.values()))
    optstate['t'] = optstate.get('t', 0) + 1
    tstep = optstate['t']
    lr_t = lr * (1 - b2 ** tstep) ** 0.5 / (1 - b1 ** tstep)
    with torch.no_grad():
        for P, grads, tag in ((G, gG, 'G'), (D, gD, 'D')):
            for (k, p), g in zip(P.items(), grads):
                m = optstate.setdefault((tag, k, 'm'), torch.zeros_like(p))
                v = optstate.setdefault((tag, k, 'v'), torch.zeros_like(p))
                m += (g - m) * (1 - b1)
                v += (g * g - v) * (1 - b2)
                p -= lr_t * m / (v.sqrt() + eps)
    return gen_total.item(), gan.item(), l1.item(), disc.item()
