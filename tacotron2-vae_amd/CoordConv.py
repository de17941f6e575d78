"""CoordConv first layer of the reference encoder (reference CoordConv.py:8-161, rank-2 branch).

`CoordConv2d` deliberately keeps the reference's odd parameter layout: it *is* a Conv2d
(dead `weight (32,1,3,3)` / `bias` that never receive gradients) and *owns* the live
`conv` with 1+3 input channels — both sets live in checkpoints."""
import torch
from torch import nn


def coord_channels(n, h, w, device, dtype=torch.float32):
    """xx varies along H (time'), yy along W (mel), both in [-1,1]; rr = dist from (0.5,0.5)."""
    xx = torch.arange(h, dtype=torch.int32, device=device).to(dtype) / (h - 1)
    yy = torch.arange(w, dtype=torch.int32, device=device).to(dtype) / (w - 1)
    xx = (xx * 2 - 1).view(1, 1, h, 1).expand(n, 1, h, w)
    yy = (yy * 2 - 1).view(1, 1, 1, w).expand(n, 1, h, w)
    rr = torch.sqrt((xx - 0.5) ** 2 + (yy - 0.5) ** 2)
    return xx, yy, rr


class AddCoords(nn.Module):
    def __init__(self, rank=2, with_r=False):
        super().__init__()
        if rank != 2:
            raise NotImplementedError("only the rank-2 branch is on the Tacotron2-VAE path")
        self.rank, self.with_r = rank, with_r

    def forward(self, x):
        n, _, h, w = x.shape
        xx, yy, rr = coord_channels(n, h, w, x.device, x.dtype)
        parts = [x, xx, yy] + ([rr] if self.with_r else [])
        return torch.cat(parts, dim=1)


class CoordConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, with_r=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.rank = 2
        self.addcoords = AddCoords(self.rank, with_r)
        self.conv = nn.Conv2d(in_channels + self.rank + int(with_r), out_channels, kernel_size, stride,
                              padding, dilation, groups, bias)

    def forward(self, x):
        """the live layer runs inside ReferenceEncoder.forward (fused CoordConv + Conv2d s2 + BatchNorm + ReLU HIP
        kernel, coordinates generated in-kernel); a direct call would be a stock-library convolution"""
        import t2v_hip
        raise t2v_hip.T2VHipError("CoordConv2d.forward is not a product path: ReferenceEncoder.forward runs the fused "
                                  "HIP kernel on conv.weight")
