"""Ulysses sequence-parallel data movement on RCCL (torch.distributed backend "nccl" == RCCL on
ROCm; xGMI point-to-point links between the 8 GPUs of a node).

The reference delegates this to un-vendored libraries: xfuser's xFuserLongContextAttention
(wan/distributed/xdit_context_parallel.py:185-190) and FastVideo's all_to_all_4D
(scripts/train/model/model_seq.py:232-234,256).  Semantics:

  seq_to_head : [L/P tokens, N heads]   -> [L tokens, N/P heads]   (q, k, v before attention)
  head_to_seq : [L tokens, N/P heads]   -> [L/P tokens, N heads]   (attention output)
  all_gather_seq : rank-order concatenation along tokens            (head output, once/forward)

All three are single collectives on contiguous buffers (all_to_all_single / all_gather_into_tensor):
on the fully connected xGMI mesh an all-to-all sends each peer its 1/P slice over its own direct
link, all 7 links concurrently.  Tokens are sharded contiguously: rank r owns [r*L/P, (r+1)*L/P).

`gloo` with CUDA tensors (the 2-process test on a 1-GPU box) is staged through host memory —
test plumbing only; production is RCCL, device to device.
"""
import torch
import torch.distributed as dist


def _a2a(recv, send, group):
    if send.is_cuda and dist.get_backend(group) == 'gloo':
        s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, s, group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group)


def seq_to_head(x, out, group, P, heads, head_dim):
    """x [Lloc, heads*hd] (row stride free) -> out [P*Lloc, (heads/P)*hd], tokens in rank order."""
    Lloc = x.shape[0]
    nl = (heads // P) * head_dim
    send = x.reshape(Lloc, P, nl).transpose(0, 1).contiguous()      # [dest rank][token][local heads]
    _a2a(out.view(P, Lloc, nl), send, group)
    return out


def head_to_seq(x, out, group, P, heads, head_dim):
    """x [P*Lloc, (heads/P)*hd] -> out [Lloc, heads*hd]."""
    nl = (heads // P) * head_dim
    Lloc = x.shape[0] // P
    recv = torch.empty(P, Lloc, nl, dtype=x.dtype, device=x.device)  # [source rank = head group][token][..]
    _a2a(recv, x.view(P, Lloc, nl), group)
    out.view(Lloc, P, nl).copy_(recv.transpose(0, 1))
    return out


def all_gather_seq(x, group, P):
    """x [Lloc, C] -> [P*Lloc, C] (rank-order concatenation; get_sp_group().all_gather(dim=1))."""
    out = torch.empty(P * x.shape[0], x.shape[1], dtype=x.dtype, device=x.device)
    if x.is_cuda and dist.get_backend(group) == 'gloo':
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(P)]
        dist.all_gather(parts, x.cpu().contiguous(), group=group)
        out.copy_(torch.cat(parts, 0))
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def p2p_send(x, dst, group=None):
    """activation of a pipeline cut -> rank dst: a 4-int64 shape header, then the fp32 payload."""
    hdr = torch.tensor(list(x.shape) + [0] * (4 - x.dim()), dtype=torch.int64)
    if dist.get_backend(group) == 'gloo':            # tests: device tensors are staged through the host
        dist.send(hdr, dst, group=group)
        dist.send(x.detach().cpu().contiguous(), dst, group=group)
    else:
        dist.send(hdr.to(x.device), dst, group=group)
        dist.send(x.contiguous(), dst, group=group)


def p2p_recv(src, device, group=None):
    gloo = dist.get_backend(group) == 'gloo'
    hdr = torch.empty(4, dtype=torch.int64, device='cpu' if gloo else device)
    dist.recv(hdr, src, group=group)
    shape = [int(v) for v in hdr.tolist() if v > 0]
    x = torch.empty(*shape, dtype=torch.float32, device='cpu' if gloo else device)
    dist.recv(x, src, group=group)
    return x.to(device)
