"""Start-up self-diagnosis of a data-parallel run (one process per GPU over RCCL; reference train.py:104-117,
util/misc.py:220-250): before the first step every rank reports which device it owns, the reports are all-gathered and
checked — so that the first multi-GPU run of a build that has only ever seen one GPU says what is wrong instead of
running slowly or hanging:

  * world size of the process group == the rank count asked for;
  * enough devices for the ranks (unless this is a rehearsal of the control flow on fewer devices);
  * ONE rank per device: no two ranks of a host report the same device identity (PCI bus id / uuid);
  * every device is a whole MI355X: 256 CUs (a compute-partitioned device — CPX / DPX — shows fewer, and the persistent
    split-K kernel sizes its grid to the CU count of a device it must have to itself, include/fvqa.h);
  * host threads are pinned per rank (OMP_NUM_THREADS set: 8 ranks with the default = all cores each fight over them).

`local_report()` touches the GPU only through torch.cuda properties; `verify()` is pure (tests/test_host_cpu.py)."""
from __future__ import annotations

import os
import socket
from typing import Dict, List

EXPECTED_CUS = 256


def local_report(rank: int, local_rank: int, device_index: int) -> Dict:
    import torch
    rep = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "device_index": device_index,
           "device_count": torch.cuda.device_count(), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
           "torch_threads": torch.get_num_threads(), "cpu_count": os.cpu_count()}
    if torch.cuda.is_available() and device_index < torch.cuda.device_count():
        p = torch.cuda.get_device_properties(device_index)
        ident = None
        for attr in ("uuid", "pci_bus_id"):
            v = getattr(p, attr, None)
            if v is not None:
                ident = f"{attr}:{v}"
                if attr == "pci_bus_id":
                    ident += f":{getattr(p, 'pci_device_id', '')}:{getattr(p, 'pci_domain_id', '')}"
                break
        rep.update(device_id=ident, cu_count=int(p.multi_processor_count), device_name=p.name,
                   gcn_arch=getattr(p, "gcnArchName", None), hbm_gib=round(p.total_memory / 2 ** 30, 1))
    return rep


def verify(reports: List[Dict], asked_world: int, group_world: int, rehearsal: bool = False) -> List[str]:
    """Problems found in the gathered reports (empty list = a sound configuration). A rehearsal (several ranks on one
    device over gloo: control flow only) may share devices and have fewer devices than ranks; everything else still holds."""
    bad = []
    if group_world != asked_world:
        bad.append(f"process group has {group_world} ranks, {asked_world} were asked for (--gpus / --nproc-per-node mismatch)")
    if len(reports) != group_world:
        bad.append(f"{len(reports)} rank reports for a group of {group_world}")
    ranks = sorted(r["rank"] for r in reports)
    if ranks != list(range(len(reports))):
        bad.append(f"rank numbers are not 0..{len(reports) - 1}: {ranks}")
    by_host: Dict[str, List[Dict]] = {}
    for r in reports:
        by_host.setdefault(r["host"], []).append(r)
    for host, rs in by_host.items():
        n_dev = min(r.get("device_count", 0) for r in rs)
        if not rehearsal and n_dev < len(rs):
            bad.append(f"host {host}: {len(rs)} ranks but {n_dev} visible devices (one process per GPU)")
        seen: Dict[str, int] = {}
        for r in rs:
            key = r.get("device_id") or f"index:{r.get('device_index')}"
            if key in seen and not rehearsal:
                bad.append(f"host {host}: ranks {seen[key]} and {r['rank']} both use device {key} "
                           "(LOCAL_RANK -> device mapping; a shared device stalls the persistent GEMM)")
            seen.setdefault(key, r["rank"])
    for r in reports:
        cu = r.get("cu_count")
        if cu is not None and cu != EXPECTED_CUS:
            bad.append(f"rank {r['rank']}: device reports {cu} CUs, a whole MI355X has {EXPECTED_CUS} "
                       "(compute partition CPX/DPX, or a CU mask?)")
        arch = r.get("gcn_arch")
        if arch is not None and "gfx950" not in str(arch):
            bad.append(f"rank {r['rank']}: device architecture {arch}, the kernels are built for gfx950")
        # host threads: what counts is the EFFECTIVE intra-op thread count of the rank (pin_host_threads caps it and exports
        # OMP_NUM_THREADS; launchers that set neither — mpirun --dist_on_itp, srun: reference util/misc.py:221-236 — are fine
        # once the rank has pinned itself). A rank whose threads are still one per core of the host is the problem.
        n_on_host = sum(1 for q in reports if q.get("host") == r.get("host"))
        thr, cpus = r.get("torch_threads"), r.get("cpu_count")
        if n_on_host > 1:
            if thr is None:
                if not r.get("omp_num_threads"):
                    bad.append(f"rank {r['rank']}: OMP_NUM_THREADS is not set ({n_on_host} ranks on {r.get('host')} would each "
                               f"start one host thread per core of {cpus})")
            elif cpus and thr * n_on_host > cpus and thr > 1:
                bad.append(f"warning: rank {r['rank']}: {thr} torch host threads x {n_on_host} ranks on {r.get('host')} oversubscribe its "
                           f"{cpus} cores (call rankcheck.pin_host_threads(world) or set OMP_NUM_THREADS per rank)")
    return bad


def check_ranks(asked_world: int, rank: int, local_rank: int, device_index: int, rehearsal: bool = False,
                strict: bool = True) -> Dict:
    """All-gather the per-rank reports, verify them on every rank, raise on rank 0 .. n (all ranks see the same list) when
    `strict`; returns {"reports": [...], "problems": [...]} for the bench line / the training log."""
    import torch.distributed as dist
    mine = local_report(rank, local_rank, device_index)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size()
        gathered: List = [None] * world
        dist.all_gather_object(gathered, mine)
    else:
        world, gathered = 1, [mine]
    problems = verify(gathered, asked_world, world, rehearsal=rehearsal)
    fatal = [p for p in problems if not p.startswith("warning:")]       # (oversubscribed host threads slow a run, they do not break it)
    if fatal and strict:
        raise RuntimeError("fvqa: the data-parallel configuration is not sound:\n  " + "\n  ".join(problems))
    if rank == 0:
        for p in problems:
            if p.startswith("warning:"):
                print(f"[fvqa.rankcheck] {p}", flush=True)
    return {"reports": gathered, "problems": problems}


def pin_host_threads(world: int, workers_per_rank: int = 0) -> int:
    """torch host threads of this rank: its share of the host's cores minus the loader workers it will start (at least 1).
    Honours an OMP_NUM_THREADS the launcher set (bench.launcher_env / torchrun set one per rank); otherwise — mpirun with
    --dist_on_itp, SLURM srun: the launchers reference util/misc.py:221-236 also accepts set none — it EXPORTS the value it
    chose, so that child processes (loader workers) and the start-up check see the same number."""
    import torch
    env = os.environ.get("OMP_NUM_THREADS")
    if env:
        n = max(1, int(env))
    else:
        n = max(1, (os.cpu_count() or 1) // max(1, world) - workers_per_rank)
        os.environ["OMP_NUM_THREADS"] = str(n)
    torch.set_num_threads(n)
    return n
