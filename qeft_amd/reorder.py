"""Runtime part of OGR: the dense index list used to gather o_proj's input (reference qeft/reorder.py:6-12)."""
import torch


def sparse_to_dense_ids(sparse_ids, length):
    """Non-outlier indices in ascending order followed by the outlier indices as given."""
    assert len(sparse_ids) < length
    keep = torch.ones(length, dtype=torch.bool, device=sparse_ids.device)
    keep[sparse_ids.long()] = False
    rest = torch.nonzero(keep, as_tuple=False).flatten()
    return torch.cat([rest, sparse_ids.to(rest.dtype)])
