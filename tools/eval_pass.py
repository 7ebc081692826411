"""Times eval-mode ResNet-152 passes (BatchNorm folded into the convolutions: the bias / ReLU / row-residual epilogues of the generic
kernels, csrc/gemm.hip EPIX 0 and 6) next to train-mode passes of the same batch.  usage: python tools/eval_pass.py [batch]"""
import sys, time
import torch
sys.path.insert(0, ".")
from situation_recognition_amd.model import resnet


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = resnet(out_layers=None, depth=152).to(dev)
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        net.eval()
        dt = timed(lambda: net(x))
        print(f"resnet152 eval : batch {B}: {dt * 1e3:8.2f} ms  ({B / dt:8.0f} img/s)")
        net.train()
        dt = timed(lambda: net(x))
        print(f"resnet152 train: batch {B}: {dt * 1e3:8.2f} ms  ({B / dt:8.0f} img/s)")


if __name__ == "__main__":
    main()
