"""reference utils/utils.py: checkpoint loading into an existing net and the metric formatter."""
import torch


def load_net(fname, net_list):
    """Key-wise copy of checkpoint['model_state_dict'] into each net's state_dict (reference utils.py:5-31).
    Works for a bare module or one wrapped in an object exposing `.module`.  The reference drops into pdb on a
    mismatch; here the error is raised."""
    checkpoint = torch.load(fname, map_location="cpu", weights_only=True)
    src = checkpoint['model_state_dict']
    for net in net_list:
        net = getattr(net, 'module', net)
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if k in src:
                    if tuple(src[k].shape) != tuple(v.shape):
                        raise RuntimeError('[Error loading] parameter[{}] size mismatch: {} vs {}'.format(k, tuple(src[k].shape), tuple(v.shape)))
                    v.copy_(src[k])
                else:
                    print('[Missed]: {}'.format(k), v.size())


def format_dict(d, s, p):
    """reference utils.py:34-42."""
    return ", ".join(p + str(k) + ": " + s.format(v * 100) for k, v in d.items())
