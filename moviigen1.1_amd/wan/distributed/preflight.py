"""First-contact check of a multi-GPU launch (`bench.py --gpus N`, scripts/inference/generate.py under torchrun): what this node and this
process group can really do, measured BEFORE the warm-up and reported in the bench line — so that a first run on an 8-GPU node says what it ran
on instead of guessing (counterpart of the reference's environment setup, wan/distributed/xdit_context_parallel.py:137-148,
185-190 + scripts/inference/generate.py:190-229, which assumes NCCL + NVSwitch and checks nothing).

    report = preflight.run(group, device, probe_peer_copy=..., budget_s=60)

Stages, each inside its own try/except (a stage that fails is recorded and the next one still runs); before every stage rank 0's clock
decides for the whole group whether the time box still allows it:
  ranks       backend, world size, every rank's (host, device index, device name, uuid)                -> rccl_ranks, rank_devices
  peer_access hipDeviceCanAccessPeer from this rank's device to every other rank's device (same host) -> peer_access
  all_to_all  PROBE_BYTES through the transport the exchange uses by default (all_to_all_single on the group), 3 repeats after 1 warm-up:
              bytes a rank sends to the OTHER ranks / the slowest rank's time                          -> link_gbps_measured.all_to_all
  peer_copy   (only when asked for: the copy-engine transport is opt-in) open IPC windows on two PROBE_BYTES buffers, pattern self-check,
              the same exchange as one-sided copies, timed the same way                                -> ipc_open, link_gbps_measured.peer_copy
The result also names the transport the measured numbers recommend (`recommended`) — a recommendation in the report, nothing switches itself.
torch.distributed is the control plane here; no tensor arithmetic."""
import os
import socket
import time

import torch
import torch.distributed as dist

from . import collectives, peer_copy

PROBE_BYTES = 64 << 20


def _gather(obj, group):
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def _go_on(t0, budget_s, group, device):
    """rank 0's clock decides for everyone (a collective must be entered by all ranks or by none)."""
    return int(collectives.control_broadcast(1 if time.perf_counter() - t0 < budget_s else 0, 0, group, device, dtype=torch.int32)) == 1


def _sync(device):
    if device.type == 'cuda':
        torch.cuda.synchronize(device)


def _timed_exchange(fn, device, group, repeats=3):
    """-> seconds per call: max over the ranks of the mean of `repeats` calls after one warm-up call."""
    fn()
    _sync(device)
    dist.barrier(group=group)
    t = time.perf_counter()
    for _ in range(repeats):
        fn()
    _sync(device)
    return float(collectives.control_reduce((time.perf_counter() - t) / repeats, 'max', group, device))


def parse(report):
    """The keys bench.py copies into its JSON line, from a report of run() (also used on a report read back from a log): ->
    dict(rccl_ranks, transport_recommended, link_gbps_measured, peer_access_all, ipc_open, preflight_s, preflight_errors)."""
    pa = report.get('peer_access')
    return {'rccl_ranks': report.get('rccl_ranks', 0),
            'transport_recommended': report.get('recommended', 'torch'),
            'link_gbps_measured': dict(report.get('link_gbps_measured') or {}),
            'peer_access_all': (all(all(v for v in row if v is not None) for row in pa) if pa else None),
            'ipc_open': report.get('ipc_open'),
            'preflight_s': report.get('elapsed_s'),
            'preflight_errors': list(report.get('errors') or [])}


def recommend(report, margin=1.05):
    """'peer_copy' only when its windows opened, its self-check passed on every rank AND it moved the probe at least `margin` x faster than the
    collective; otherwise 'torch' (the collective).  Any recorded error of the peer-copy stage means 'torch'."""
    g = report.get('link_gbps_measured') or {}
    if report.get('ipc_open') is True and g.get('peer_copy') and g.get('all_to_all') and not any('peer_copy' in e for e in report.get('errors') or []):
        if g['peer_copy'] >= margin * g['all_to_all']:
            return 'peer_copy'
    return 'torch'


def run(group, device, probe_peer_copy=False, budget_s=60.0, probe_bytes=PROBE_BYTES):
    group = group if group is not None else dist.group.WORLD
    device = torch.device(device)
    P, r = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    t0 = time.perf_counter()
    rep = {'backend': backend, 'world': P, 'rccl_ranks': P if backend == 'nccl' else 0, 'probe_bytes': probe_bytes, 'errors': [],
           'link_gbps_measured': {}, 'ipc_open': None, 'peer_access': None, 'stages_skipped': []}

    # ---- ranks ----------------------------------------------------------------------------------------------------------------
    try:
        if device.type == 'cuda':
            props = torch.cuda.get_device_properties(device)
            mine = {'rank': r, 'host': socket.gethostname(), 'device': device.index, 'name': props.name, 'uuid': str(getattr(props, 'uuid', '')),
                    'visible': torch.cuda.device_count(), 'ipc_mode_legacy': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}
        else:       # host tensors (the CPU tests of this file's logic): no device to describe, no peer access to ask for
            mine = {'rank': r, 'host': socket.gethostname(), 'name': 'cpu'}
    except Exception as e:      # noqa: BLE001
        mine = {'rank': r, 'error': f'{type(e).__name__}: {e}'}
    try:
        rep['rank_devices'] = _gather(mine, group)
    except Exception as e:      # noqa: BLE001
        rep['errors'].append(f'ranks: {type(e).__name__}: {e}')
        rep['rank_devices'] = [mine]

    # ---- peer access (no communication: a driver query per peer) ---------------------------------------------------------------------
    try:
        row = []
        for d in rep['rank_devices']:
            if d is None or 'device' not in d or d.get('host') != mine.get('host'):
                row.append(None)
            elif d['device'] == device.index:
                row.append(True)          # the same device (this rank, or ranks sharing a GPU in the gloo tests)
            elif d['device'] < torch.cuda.device_count():
                row.append(bool(torch.cuda.can_device_access_peer(device.index, d['device'])))
            else:
                row.append(None)          # not visible to this process (a per-rank visibility mask)
        rep['peer_access'] = _gather(row, group)
    except Exception as e:      # noqa: BLE001
        rep['errors'].append(f'peer_access: {type(e).__name__}: {e}')

    # ---- the collective the exchange uses by default ---------------------------------------------------------------------------------
    n = max(P, probe_bytes // 2 // P * P)             # bf16 elements, a multiple of P
    sent_to_others = n * 2 * (P - 1) / P
    if P > 1 and _go_on(t0, budget_s, group, device):
        try:
            send = torch.empty(P, n // P, dtype=torch.bfloat16, device=device).fill_(float(r))
            recv = torch.empty_like(send)
            dt = _timed_exchange(lambda: collectives.all_to_all(recv, send, group), device, group)
            ok = all(bool((recv[p] == float(p)).all()) for p in range(P))
            rep['link_gbps_measured']['all_to_all'] = sent_to_others / dt / 1e9
            if not ok:
                rep['errors'].append('all_to_all: the probe pattern did not arrive intact')
        except Exception as e:      # noqa: BLE001
            rep['errors'].append(f'all_to_all: {type(e).__name__}: {e}')
    elif P > 1:
        rep['stages_skipped'].append('all_to_all')

    # ---- the copy-engine transport, only when asked for ------------------------------------------------------------------------------
    if P > 1 and probe_peer_copy:
        if _go_on(t0, budget_s, group, device):
            try:
                bufs = [torch.empty(P, n // P, dtype=torch.bfloat16, device=device) for _ in range(2)]
                win = peer_copy.open_windows(group, bufs)           # all ranks or none; never raises past its vote
                rep['ipc_open'] = win is not None
                if win is not None:
                    send = torch.empty(P, n // P, dtype=torch.bfloat16, device=device).fill_(float(r))
                    dt = _timed_exchange(lambda: win.all_to_all(0, send), device, group)
                    ok = all(bool((bufs[0][p] == float(p)).all()) for p in range(P)) and not win.failed()
                    rep['link_gbps_measured']['peer_copy'] = sent_to_others / dt / 1e9
                    if not ok:
                        rep['errors'].append('peer_copy: the probe pattern did not arrive intact')
                    del win
            except Exception as e:      # noqa: BLE001
                rep['errors'].append(f'peer_copy: {type(e).__name__}: {e}')
                rep['ipc_open'] = False if rep['ipc_open'] is None else rep['ipc_open']
        else:
            rep['stages_skipped'].append('peer_copy')

    rep['elapsed_s'] = time.perf_counter() - t0
    rep['recommended'] = recommend(rep)
    return rep
