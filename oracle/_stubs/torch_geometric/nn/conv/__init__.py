"""Stand-in for `torch_geometric.nn.conv.MessagePassing` (test infrastructure only).

Implements the subset of the published PyG contract that the reference's LEFTNet
layers rely on: `propagate(edge_index, size=None, **kwargs)` gathers every
`<name>_j` argument from `kwargs[name][edge_index[0]]` and every `<name>_i`
from `kwargs[name][edge_index[1]]` (flow = source_to_target), calls
`message`, then `aggregate(inputs, index=edge_index[1], ptr=None, dim_size=N)`
and `update`.
"""
import inspect

import torch
from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim

    def jittable(self, *args, **kwargs):
        return self

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        dim_size = None
        for v in kwargs.values():
            if isinstance(v, torch.Tensor):
                dim_size = v.size(self.node_dim)
                break
        params = list(inspect.signature(self.message).parameters)
        msg_kwargs = {}
        for name in params:
            if name.endswith("_j"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(self.node_dim, src)
            elif name.endswith("_i"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(self.node_dim, dst)
            else:
                msg_kwargs[name] = kwargs[name]
        out = self.message(**msg_kwargs)
        out = self.aggregate(out, dst, None, dim_size)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        reduce = "sum" if self.aggr == "add" else self.aggr
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=reduce)

    def update(self, inputs):
        return inputs
