"""Deterministic integer-hash evaluator that lives on the GPU.

Not a model: a pure function of (bitboards, side to move) built from 64-bit integer mixing,
small integers and one correctly rounded fp32 division, so that numpy on the host, torch on
the device and a HIP kernel all produce bit-identical policy / WDL / moves-left values
(tests/scenarios.py holds the numpy twin).  Used to check the fused search path bit-for-bit
against the CPU oracle and to benchmark the tree kernels without a network.
"""
import numpy as np
import torch


class NumpyHashEvaluator:
    """The same function on the host in numpy, behind the reference's `predict(state, action_mask)`
    contract (Connect4/Network.py:267-288): the evaluator of bench.py's tree-only CPU baseline."""
    n_actions = 7

    def predict(self, state, action_mask=None):
        state = np.asarray(state)
        turns = state[:, 2, 0, 0].astype(np.int64)
        grid = ((state[:, 0] - state[:, 1]) * turns[:, None, None]).astype(np.int8)
        rows, cols = np.arange(6).reshape(1, 6, 1), np.arange(7).reshape(1, 1, 7)
        bit = (np.uint64(1) << (cols * 7 + (5 - rows)).astype(np.uint64))
        bb0 = (bit * (grid == 1)).sum((1, 2), dtype=np.uint64)
        bb1 = (bit * (grid == -1)).sum((1, 2), dtype=np.uint64)
        with np.errstate(over="ignore"):
            x = bb0 * np.uint64(0x9E3779B97F4A7C15)
            x ^= (bb1 + np.uint64(0x7F4A7C159E3779B9)) * np.uint64(0xBF58476D1CE4E5B9)
            x += np.where(turns == 1, np.uint64(0x94D049BB133111EB), np.uint64(0x2545F4914F6CDD1D))
            x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
            x ^= x >> np.uint64(31)
        sh = (np.arange(7) * 4).astype(np.uint64)
        probs = (1 + ((x[:, None] >> sh) & np.uint64(15))).astype(np.float32) / np.float32(16)
        if action_mask is not None:
            probs = probs * np.asarray(action_mask, dtype=np.float32)
        w = np.stack([1 + ((x >> np.uint64(s)) & np.uint64(31)) for s in (28, 33, 38)], 1)
        wdl = w.astype(np.float32) / w.sum(1, keepdims=True).astype(np.float32)
        ml = ((x >> np.uint64(43)) & np.uint64(63)).astype(np.float32) / np.float32(2)
        return probs, wdl.astype(np.float32), ml.reshape(-1, 1)


def _c(v):
    """uint64 constant as the int64 bit pattern torch computes with."""
    return v - (1 << 64) if v >= (1 << 63) else v


def _xorshift(x, s):
    return x ^ ((x >> s) & ((1 << (64 - s)) - 1))      # logical shift on int64


class HashEvaluator(torch.nn.Module):
    """Follows the device-evaluator protocol of fused.FusedSearch: `predict_device(features,
    mask) -> (probs[n,7], wdl_rel[n,3], moves_left[n])`, all float32 tensors on the device."""
    is_device_evaluator = True
    n_actions = 7
    native_hash_game = 0        # AZ_GAME_CONNECT4: az_nn_model_create_hash computes the same function

    def __init__(self, device="cuda"):
        super().__init__()
        self.register_buffer("anchor", torch.zeros(1, device=device))

    @torch.no_grad()
    def predict_device(self, feats, mask):
        dev = feats.device
        turn = feats[:, 2, 0, 0].to(torch.int64)
        grid = ((feats[:, 0] - feats[:, 1]) * feats[:, 2]).to(torch.int64)         # +1 / -1 / 0
        rows = torch.arange(6, device=dev).view(1, 6, 1)
        cols = torch.arange(7, device=dev).view(1, 1, 7)
        bit = torch.ones((), dtype=torch.int64, device=dev) << (cols * 7 + (5 - rows))
        bb0 = (bit * (grid == 1)).sum((1, 2))
        bb1 = (bit * (grid == -1)).sum((1, 2))
        x = bb0 * _c(0x9E3779B97F4A7C15)
        x = x ^ ((bb1 + _c(0x7F4A7C159E3779B9)) * _c(0xBF58476D1CE4E5B9))
        x = x + torch.where(turn == 1, torch.full_like(x, _c(0x94D049BB133111EB)),
                            torch.full_like(x, _c(0x2545F4914F6CDD1D)))
        x = _xorshift(x, 30) * _c(0xBF58476D1CE4E5B9)
        x = _xorshift(x, 27) * _c(0x94D049BB133111EB)
        x = _xorshift(x, 31)
        sh = torch.arange(7, device=dev) * 4
        probs = (1 + ((x.unsqueeze(1) >> sh) & 15)).to(torch.float32) / 16.0
        probs = probs * mask.to(torch.float32)
        w = torch.stack([1 + ((x >> s) & 31) for s in (28, 33, 38)], 1)
        wdl = w.to(torch.float32) / w.sum(1, keepdim=True).to(torch.float32)
        ml = ((x >> 43) & 63).to(torch.float32) / 2.0
        return probs, wdl, ml

    @torch.no_grad()
    def predict(self, state, action_mask=None):
        """numpy contract of the reference's `predict`, for the host path."""
        import numpy as np
        f = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.anchor.device)
        m = torch.ones((f.shape[0], 7), dtype=torch.bool, device=f.device) if action_mask is None \
            else torch.as_tensor(np.asarray(action_mask), device=f.device).to(torch.bool)
        p, w, ml = self.predict_device(f, m)
        return p.cpu().numpy(), w.cpu().numpy(), ml.view(-1, 1).cpu().numpy()


class OthelloHashEvaluator(torch.nn.Module):
    """The same idea for Othello (tests/scenarios.py ot_hash_eval): 65 policy values from five
    re-mixed words, WDL from three 5-bit weights, auxiliary utility = 6 bits / 32 - 1."""
    is_device_evaluator = True
    n_actions = 65
    native_hash_game = 1        # AZ_GAME_OTHELLO

    def __init__(self, device="cuda"):
        super().__init__()
        self.register_buffer("anchor", torch.zeros(1, device=device))

    @torch.no_grad()
    def predict_device(self, feats, mask):
        dev = feats.device
        n = feats.shape[0]
        turn = feats[:, 2, 0, 0].to(torch.int64)
        grid = ((feats[:, 0] - feats[:, 1]) * feats[:, 2]).to(torch.int64).reshape(n, 64)
        bit = torch.ones((), dtype=torch.int64, device=dev) << torch.arange(64, device=dev)
        bb0 = (bit * (grid == 1)).sum(1)
        bb1 = (bit * (grid == -1)).sum(1)
        x = bb0 * _c(0x9E3779B97F4A7C15)
        x = x ^ ((bb1 + _c(0x7F4A7C159E3779B9)) * _c(0xBF58476D1CE4E5B9))
        x = x + torch.where(turn == 1, torch.full_like(x, _c(0x94D049BB133111EB)),
                            torch.full_like(x, _c(0x2545F4914F6CDD1D)))
        x = _xorshift(x, 30) * _c(0xBF58476D1CE4E5B9)
        x = _xorshift(x, 27) * _c(0x94D049BB133111EB)
        x = _xorshift(x, 31)
        sh = torch.arange(16, device=dev) * 4
        words = []
        for k in range(5):
            hk = x + _c((0x9E3779B97F4A7C15 * (k + 1)) & ((1 << 64) - 1))
            hk = _xorshift(hk, 29) * _c(0xBF58476D1CE4E5B9)
            hk = _xorshift(hk, 32)
            words.append((1 + ((hk.unsqueeze(1) >> sh) & 15)).to(torch.float32) / 16.0)
        probs = torch.cat(words, 1)[:, :65] * mask.to(torch.float32)
        w = torch.stack([1 + ((x >> s) & 31) for s in (28, 33, 38)], 1)
        wdl = w.to(torch.float32) / w.sum(1, keepdim=True).to(torch.float32)
        aux = ((x >> 43) & 63).to(torch.float32) / 32.0 - 1.0
        return probs, wdl, aux

    @torch.no_grad()
    def predict(self, state, action_mask=None):
        import numpy as np
        f = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.anchor.device)
        m = torch.ones((f.shape[0], 65), dtype=torch.bool, device=f.device) if action_mask is None \
            else torch.as_tensor(np.asarray(action_mask), device=f.device).to(torch.bool)
        p, w, aux = self.predict_device(f, m)
        return p.cpu().numpy(), w.cpu().numpy(), aux.view(-1, 1).cpu().numpy()
