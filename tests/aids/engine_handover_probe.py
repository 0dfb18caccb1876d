#!/usr/bin/env python
"""Does the autograd engine announce a gradient that crosses streams to the caching allocator (record_stream on the consumer's
stream)?  A gradient produced on the main stream is consumed by a node whose forward ran on a side stream that is kept busy; right
after backward() returns, an allocation of the gradient's size on the main stream must NOT get the gradient's block back."""
import torch

dev = torch.device("cuda", 0)
side = torch.cuda.Stream()
n = 1 << 20
seen = {}


class OnSide(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        seen["ptr"], seen["stream"] = g.data_ptr(), torch.cuda.current_stream().cuda_stream
        torch.cuda._sleep(200000000)            # the consumer's stream stays busy long after backward() has returned
        return g * 2.0


class OnMain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        return y.sum()

    @staticmethod
    def backward(ctx, g):
        return torch.full((n,), 3.0, device=dev) * g        # a fresh block from the main stream's pool


x = torch.ones(n, device=dev, requires_grad=True)
torch.cuda.synchronize()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    y = OnSide.apply(x)
torch.cuda.current_stream().wait_stream(side)
loss = OnMain.apply(y)
with torch.autograd.set_multithreading_enabled(False):
    loss.backward()
again = [torch.empty(n, device=dev) for _ in range(6)]        # (the first ones take the temporaries of OnMain.backward)
hit = any(t.data_ptr() == seen["ptr"] for t in again)
print("consumer node ran on the side stream:", seen["stream"] == side.cuda_stream)
print("gradient block %#x, next main-stream allocations of that size %s -> %s" % (
    seen["ptr"], [hex(t.data_ptr()) for t in again], "REUSED AT ONCE (not announced)" if hit else "held back (record_stream was called)"))
torch.cuda.synchronize()
