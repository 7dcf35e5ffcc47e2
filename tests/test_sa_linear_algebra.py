"""CPU, float64: the closed forms csrc/sa_last_bwd.hip implements, against autograd through the reference's arithmetic
(Conv -> BatchNorm(batch statistics) -> ReLU -> max-pool over nsample; pytorch_utils.py:11-36, pointnet2_modules.py:243-257).

    dH  = [arg-max rows: g s W3[c,:]] - H A + d,      A = W3^T diag(s m2 rstd) W3,  d = W3^T (s m2 rstd mu - s m1)
    dW3 = s (T - m1 S^T - m2 rstd (W3 Gram - mu S^T))
    dW1 = s1 (GX - m1 SX^T - m2 rstd1 (W1 XX - mu1 SX^T))            (first layer, no input gradient)
"""
import torch


def test_last_layer_and_first_layer_closed_forms_equal_autograd():
    torch.manual_seed(0)
    dt = torch.float64
    G, ns, Cin, C1, C2, C3 = 37, 16, 8, 10, 12, 20
    P = G * ns
    eps = 1e-5
    X = torch.randn(P, Cin, dtype=dt)
    W1 = (torch.randn(C1, Cin, dtype=dt) / 2).requires_grad_(True)
    W2 = (torch.randn(C2, C1, dtype=dt) / 3).requires_grad_(True)
    W3 = (torch.randn(C3, C2, dtype=dt) / 3).requires_grad_(True)
    gam = [torch.randn(c, dtype=dt).requires_grad_(True) for c in (C1, C2, C3)]       # negative scales included
    bet = [(torch.randn(c, dtype=dt) * 0.3).requires_grad_(True) for c in (C1, C2, C3)]

    def layer(x, W, g, b):
        z = x @ W.T
        mu, var = z.mean(0), z.var(0, unbiased=False)
        rstd = 1 / torch.sqrt(var + eps)
        return z, mu, rstd, torch.relu((z - mu) * rstd * g + b)

    z1, mu1, rstd1, h1 = layer(X, W1, gam[0], bet[0])
    z2, mu2, rstd2, h2 = layer(h1, W2, gam[1], bet[1])
    h2.retain_grad(); h1.retain_grad()
    z3, mu3, rstd3, h3 = layer(h2, W3, gam[2], bet[2])
    pooled, arg = h3.view(G, ns, C3).max(1)
    dP = torch.randn(G, C3, dtype=dt)
    (pooled * dP).sum().backward()
    with torch.no_grad():
        # ---- last layer
        s = gam[2] * rstd3
        zsel = z3.view(G, ns, C3).gather(1, arg[:, None, :])[:, 0]
        g = dP * (((zsel - mu3) * s + bet[2]) > 0)
        zh = (zsel - mu3) * rstd3
        m1, m2 = g.sum(0) / P, (g * zh).sum(0) / P
        H, W = h2.detach(), W3.detach()
        Gram, S = H.T @ H, H.sum(0)
        rows = torch.arange(G)[:, None] * ns + arg
        T = torch.stack([(g[:, c, None] * H[rows[:, c]]).sum(0) for c in range(C3)])
        dW3 = s[:, None] * (T - m1[:, None] * S[None] - (m2 * rstd3)[:, None] * (W @ Gram - mu3[:, None] * S[None]))
        u = s * m2 * rstd3
        A, d = W.T @ (u[:, None] * W), W.T @ (u * mu3 - s * m1)
        dH = -H @ A + d[None]
        for c in range(C3):
            dH.index_add_(0, rows[:, c], (g[:, c] * s[c])[:, None] * W[c][None])
        assert float((dW3 - W3.grad).abs().max()) < 1e-10 * float(W3.grad.abs().max())
        assert float((dH - h2.grad).abs().max()) < 1e-10 * float(h2.grad.abs().max())
        assert float(((g * zh).sum(0) - gam[2].grad).abs().max()) < 1e-10 and float((g.sum(0) - bet[2].grad).abs().max()) < 1e-10
        # ---- first layer from the gradient of its activation (what butd_sa_first_bwd / _mid_first_bwd fold)
        s1 = gam[0] * rstd1
        g1 = h1.grad * (h1 > 0)
        zh1 = (z1 - mu1) * rstd1
        a1, a2 = g1.sum(0) / P, (g1 * zh1).sum(0) / P
        GX, SX, XX = g1.T @ X, X.sum(0), X.T @ X
        W1d = W1.detach()
        dW1 = s1[:, None] * (GX - a1[:, None] * SX[None] - (a2 * rstd1)[:, None] * (W1d @ XX - (W1d @ SX / P)[:, None] * SX[None]))
        assert float((dW1 - W1.grad).abs().max()) < 1e-10 * float(W1.grad.abs().max())
        assert float((g1.sum(0) - bet[0].grad).abs().max()) < 1e-10 and float(((g1 * zh1).sum(0) - gam[0].grad).abs().max()) < 1e-10
