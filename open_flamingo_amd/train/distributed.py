"""Rank discovery + process-group bootstrap: one process per GPU, RCCL (backend string "nccl" on ROCm) over xGMI.

Same entry points as the reference ``open_flamingo/train/distributed.py`` (world_info_from_env :48-70,
init_distributed_device :73-132) restricted to what this build uses: torchrun / SLURM / OpenMPI environment
variables and the env:// rendezvous.  The horovod branch of the reference is dead code there (never attached to an
optimizer) and is not reproduced."""
import os

import torch
import torch.distributed as dist

_LOCAL = ("LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "OMPI_COMM_WORLD_LOCAL_RANK")
_GLOBAL = ("RANK", "PMI_RANK", "SLURM_PROCID", "OMPI_COMM_WORLD_RANK")
_WORLD = ("WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS", "OMPI_COMM_WORLD_SIZE")


def _first_env(names, default):
    for n in names:
        if n in os.environ:
            return int(os.environ[n])
    return default


def world_info_from_env():
    """(local_rank, global_rank, world_size) from the launcher's environment."""
    return _first_env(_LOCAL, 0), _first_env(_GLOBAL, 0), _first_env(_WORLD, 1)


def is_using_distributed():
    return _first_env(_WORLD, 1) > 1


def init_distributed_device(args=None, backend=None):
    """Initialise torch.distributed (if WORLD_SIZE > 1) and pick this rank's device.  Fills
    args.{distributed,world_size,rank,local_rank,device} when an args namespace is given; returns the device."""
    local_rank, rank, world = world_info_from_env()
    distributed = world > 1
    use_gpu = torch.cuda.is_available()
    if distributed and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
        if use_gpu:
            torch.cuda.set_device(local_rank)
        be = backend or (getattr(args, "dist_backend", None) if args is not None else None) or ("nccl" if use_gpu else "gloo")
        kw = {}
        if use_gpu and be == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=be, init_method="env://", world_size=world, rank=rank, **kw)
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if args is not None:
        args.distributed, args.world_size, args.rank, args.local_rank = distributed, world, rank, local_rank
        args.device = str(device)
    return device
