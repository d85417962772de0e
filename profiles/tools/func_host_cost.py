"""Host cost of ONE evaluation of config 2's func (`y @ A.T`, 65536 x 128 fp32) as the framework dispatches it — what the solver pays per
stage on the host, whatever this library does.  Back-to-back calls without synchronisation: host time per call vs GPU time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from tests import problems as P  # noqa: E402


def measure(label, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("{:60s} host {:6.1f} us / call   wall {:6.1f} us / call".format(label, 1e6 * host / n, 1e6 * wall / n))


for B, D in ((65536, 128), (65536, 64), (1024, 128)):
    A = P.skew_matrix(D).float().cuda()
    AT = A.T.contiguous()
    y = torch.randn(B, D, device="cuda")
    out = torch.empty_like(y)
    keep = []
    with torch.no_grad():
        measure("{} x {}: y @ A.T (fresh result each call, dropped)".format(B, D), lambda: y @ A.T)
        measure("{} x {}: y @ AT (pre-transposed contiguous)".format(B, D), lambda: y @ AT)
        measure("{} x {}: torch.matmul(y, A.T, out=buf)".format(B, D), lambda: torch.matmul(y, A.T, out=out))
        measure("{} x {}: torch.empty_like(y) alone".format(B, D), lambda: torch.empty_like(y))

        def six():
            del keep[:]
            for _ in range(6):
                keep.append(y @ A.T)

        measure("{} x {}: six results kept alive, then dropped (a step's pattern), per 6".format(B, D), six, n=100)
if "--tunable" in sys.argv:
    pass
