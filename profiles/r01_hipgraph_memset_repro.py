#!/usr/bin/env python3
"""Pure-PyTorch reproduction (no paddlexde_amd involved) of the ROCm 7.2 / torch 2.10 behaviour DESIGN.md §5 (iii) describes:
a captured graph that holds a hipGraph MEMSET node — x.sum(0) of an [8192, 50] tensor captures as [memset, kernel] — returns
stale results when replayed among ordinary stream work; graphs of kernel / memcpy nodes do not.

    python profiles/r01_hipgraph_memset_repro.py        (on an MI355X box; output in r01_hipgraph_memset_repro.log)
"""
import ctypes as C

import torch

dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")


def node_types(g):
    raw, n = g.raw_cuda_graph(), C.c_size_t(0)
    hip.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n))
    arr = (C.c_void_p * n.value)()
    hip.hipGraphGetNodes(C.c_void_p(raw), arr, C.byref(n))
    out = []
    for i in range(n.value):
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(arr[i]), C.byref(t))
        out.append({0: "kernel", 1: "memcpy", 2: "memset"}.get(t.value, t.value))
    return out


def run(name, build, after_replay=None, n=300):
    x = torch.randn(8192, 50, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            build(x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        out = build(x)
    types = node_types(g)
    g.instantiate()
    junk, bad = torch.randn(8192, 64, device=dev), 0
    for _ in range(n):
        x.copy_(torch.randn(8192, 50, device=dev))
        g.replay()
        if after_replay:
            after_replay()
        got = out.clone()
        junk.sum(1)  # ordinary stream work
        bad += int(not torch.equal(got, build(x)))
    print("{:<44} nodes {:<22} stale results: {} of {}".format(name, str(types), bad, n))


buf = torch.empty(8192, 50, device=dev)
run("x.sum(0)   (multi-block reduction)", lambda x: x.sum(0))
run("x.sum(0) + event.synchronize()", lambda x: x.sum(0), lambda: (lambda e: (e.record(), e.synchronize()))(torch.cuda.Event()))
run("x.sum(0) + stream.synchronize()", lambda x: x.sum(0), lambda: torch.cuda.current_stream().synchronize())
run("x.sum(1)   (single-block reduction)", lambda x: x.sum(1))
run("buf.copy_(x) * 2   (memcpy node)", lambda x: buf.copy_(x) * 2)
run("(x * 2 + 1).tanh()", lambda x: (x * 2 + 1).tanh())
