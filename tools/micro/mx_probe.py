#!/usr/bin/env python3
"""drives tools/micro/mx_probe (layout discovery of the MX-scaled fp8 MFMA): random e4m3 A [16][128], B [16][128] (= B^T),
runs the four layout hypotheses on the GPU and says which reproduces A . B^T of the dequantised values"""
import os, subprocess, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(1)
A = (torch.randn(16, 128, generator=g) * 0.7).to(torch.float8_e4m3fn)
B = (torch.randn(16, 128, generator=g) * 0.7).to(torch.float8_e4m3fn)
open("/tmp/mxA.bin", "wb").write(A.view(torch.uint8).numpy().tobytes())
open("/tmp/mxB.bin", "wb").write(B.view(torch.uint8).numpy().tobytes())
ref = A.float() @ B.float().t()
out = subprocess.run([os.path.join(here, "mx_probe"), "layout", "/tmp/mxA.bin", "/tmp/mxB.bin"], capture_output=True, text=True).stdout
for line in out.strip().split("\n"):
    tok = line.split()
    C = torch.tensor([float(x) for x in tok[1:]]).view(16, 16)
    print(tok[0], "max |C - A.B^T| = %.3e" % (C - ref).abs().max().item(), " max |C - (A.B^T)^T| = %.3e" % (C - ref.t()).abs().max().item())
