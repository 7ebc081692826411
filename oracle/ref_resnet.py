"""CPU restatement of the torchvision ResNet (v1.5) graph the reference builds at
model.py:16 (`tv.models.resnet152`) and runs at model.py:35.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the
reference: torchvision is a third-party dependency with no pinned version
(reference README.md:11 says only "PyTorch 1.6+"), absent from /root/reference
and from this image.  What is restated here is the published architecture:

  conv 7x7/2 p3 (3->w) -> BN -> ReLU -> maxpool 3x3/2 p1 ->
  4 stages of residual blocks (stage s: planes = w*2^s, first block of stages
  1..3 has stride 2, the stride sits on the 3x3 conv = "v1.5") ->
  global average pool -> flatten -> fc.

  Bottleneck: 1x1 -> BN -> ReLU -> 3x3(stride) -> BN -> ReLU -> 1x1(x4) -> BN,
              (+ 1x1(stride) conv + BN on the identity when shape changes), add, ReLU.
  BasicBlock: 3x3(stride) -> BN -> ReLU -> 3x3 -> BN, (+ downsample), add, ReLU.

BatchNorm eps 1e-5, momentum 0.1, no conv bias; parameter/buffer names equal
torchvision's (`conv1.weight`, `layer3.7.bn2.running_var`,
`layer2.0.downsample.0.weight`, ...), so a reference checkpoint's
`convnet_verbs.model.*` keys line up.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

ARCH = {
    # depth: (block kind, blocks per stage)
    18: ("basic", (2, 2, 2, 2)),
    34: ("basic", (3, 4, 6, 3)),
    50: ("bottleneck", (3, 4, 6, 3)),
    101: ("bottleneck", (3, 4, 23, 3)),
    152: ("bottleneck", (3, 8, 36, 3)),
}


class _Downsample(nn.Sequential):
    def __init__(self, cin, cout, stride):
        super().__init__(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                         nn.BatchNorm2d(cout))


class RefBottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        cout = planes * self.expansion
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = _Downsample(cin, cout, stride) if (stride != 1 or cin != cout) else None

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idn)


class RefBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = _Downsample(cin, planes, stride) if (stride != 1 or cin != planes) else None

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + idn)


class RefResNet(nn.Module):
    """`RefResNet(152)` is the graph of torchvision.models.resnet152; `width`
    (64 in torchvision) and `blocks` can be shrunk for KB-sized test fixtures."""

    def __init__(self, depth=152, width=64, blocks=None, num_classes=1000):
        super().__init__()
        kind, default_blocks = ARCH[depth]
        blocks = tuple(blocks) if blocks is not None else default_blocks
        Block = RefBottleneck if kind == "bottleneck" else RefBasicBlock
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        cin = width
        for s in range(4):
            planes = width << s
            stage = []
            for i in range(blocks[s]):
                stage.append(Block(cin, planes, 2 if (i == 0 and s > 0) else 1))
                cin = planes * Block.expansion
            setattr(self, "layer%d" % (s + 1), nn.Sequential(*stage))
        self.fc = nn.Linear(cin, num_classes)
        # torchvision's initialisation: He-normal (fan_out) convs, BN gamma=1 beta=0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)


def perturb_batchnorm_(net, seed):
    """Give every BN non-trivial gamma/beta/running stats (a freshly initialised
    net has gamma=1, beta=0, mean=0, var=1, which hides scale/shift bugs)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return net


def calibrate_batchnorm_(net, img):
    """Set every BatchNorm's running statistics to the batch statistics of `img` (one train-mode pass with momentum 1).
    A He-initialised net with arbitrary running statistics is not normalised in eval mode: activations grow by orders of
    magnitude through 50 residual blocks and absolute tolerances on its logits mean nothing.  A TRAINED net's running
    statistics are the statistics of its activations -- this puts a randomly initialised test net into that regime."""
    bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    keep = [(m.momentum, m.training) for m in bns]
    was = net.training
    net.train()
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        net(img)
    for m, (mom, _) in zip(bns, keep):
        m.momentum = mom
        m.num_batches_tracked.zero_()
    net.train(was)
    return net
