"""Checkpoint conventions of the reference's trainers (core/gcn_trainer.py:90-100, 298-311; core/lijun_trainer.py:343-355), so
that files written by the reference load here and vice versa: a file is either a bare `state_dict` or
`{'epoch': e, 'network': state_dict}`, and its keys may carry DistributedDataParallel's `module.` prefix (the reference strips
the first 7 characters of every key when the first `load_state_dict` fails).  The module trees of this package reproduce the
reference's parameter names (tests/test_cpu_host.py::test_state_dict_schema_*), which is what makes this a plain
`load_state_dict`."""
import torch


def load_checkpoint(network, path, map_location='cpu'):
    """Returns the epoch stored in the file (None for a bare state dict)."""
    state = torch.load(path, map_location=map_location)
    epoch = None
    if isinstance(state, dict) and 'network' in state:
        epoch = state.get('epoch')
        state = state['network']
    try:
        network.load_state_dict(state)
    except RuntimeError:
        network.load_state_dict({k[7:]: v for k, v in state.items()})        # 'module.' (gcn_trainer.py:96-100)
    refresh = [m for m in (network, getattr(network, 'encoder', None)) if m is not None]
    for m in refresh:                                                          # inference snapshots follow the weights
        if getattr(m, '_half', None) is not None:
            m.use_fp16_backbone()
        if getattr(m, '_folded', None) is not None:
            m.fold_batchnorm()
    return epoch


def save_checkpoint(network, path, epoch):
    """`{'epoch', 'network'}` with the DDP wrapper stripped, as core/gcn_trainer.py:303-311."""
    net = getattr(network, 'module', network)
    torch.save({'epoch': epoch, 'network': net.state_dict()}, path)
