"""The NN evaluation seam — drop-in for reversi_zero/agent/api.py:20-45.

`ReversiModelAPI.predict(x)` keeps the reference's contract: x is (N,2,8,8) or (2,8,8) planes
[own, enemy] of the side to move (agent/player.py:307-309); returns (policy (N,64), value (N,1)) as
float32 numpy arrays, or (policy (64,), value (1,)) for a single position.  The planes are packed into
bitboards and evaluated by the HIP forward pass (`raz_net_forward`, include/raz.h) — the same kernels
the batched engine uses for its cross-game leaf batches.  The reference's Pipe server/client pair
(api.py:48-141) has no counterpart: all games live on the device, there is nothing to fan in.
"""
import numpy as np


class ReversiModelAPI:
    def __init__(self, config, agent_model, device="cuda:0"):
        """agent_model: reversi_alpha_zero_amd.agent.model.ReversiModel (its `.model` is the torch
        restatement of the Keras graph) or anything with `to_blob()`."""
        self.config = config
        self.agent_model = agent_model
        self.device = device
        self._net = None

    def _device_net(self):
        if self._net is None:
            from ..engine import DeviceNet
            m = getattr(self.agent_model, "model", self.agent_model)
            self._net = DeviceNet(m.to_blob(), self.device)
        return self._net

    def predict(self, x):
        x = np.asarray(x)
        assert x.ndim in (3, 4)
        assert x.shape == (2, 8, 8) or x.shape[1:] == (2, 8, 8)
        orig = x
        if x.ndim == 3:
            x = x.reshape(1, 2, 8, 8)
        policy, value = self._do_predict(x)
        if orig.ndim == 3:
            return policy[0], value[0]
        return policy, value

    def _do_predict(self, x):
        import torch
        net = self._device_net()
        bits = (x.reshape(x.shape[0], 2, 64) != 0).astype(np.uint64)
        packed = (bits << np.arange(64, dtype=np.uint64)[None, None, :]).sum(axis=2, dtype=np.uint64)
        own = torch.from_numpy(packed[:, 0].copy().view(np.int64)).to(net.device)
        enemy = torch.from_numpy(packed[:, 1].copy().view(np.int64)).to(net.device)
        pol, val = net.predict_bitboards(own, enemy)
        return pol.cpu().numpy(), val.cpu().numpy().reshape(-1, 1)
