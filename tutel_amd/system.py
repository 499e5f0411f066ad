"""tutel.system facade (reference: tutel/system.py:27-79): session bootstrap + timing."""
import atexit
import logging
import os
import re
import sys
import time


def init_data_model_parallel(group_count=1, backend="nccl"):
    """torch.distributed bootstrap (backend "nccl" is RCCL on ROCm, "gloo" on CPU) and the
    data/model group split; returns the DistributedProperties bag."""
    from . import net
    env = net.create_groups_from_world(group_count=group_count, include_init=backend)
    env.is_cuda = env.local_device is not None and env.local_device.type == "cuda"
    logging.critical(f"Registering device global rank {env.global_rank}: data_rank = {env.data_rank}, model_rank = {env.model_rank}")
    init_data_model_parallel.default_env = env

    def _quit():
        sys.stdout.flush()
        sys.stderr.flush()
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass

    atexit.register(_quit)
    return env


def get_local_session():
    if not hasattr(init_data_model_parallel, "default_env"):
        raise Exception("Current session is not initialized with: system.init_data_model_parallel() from tutel. Please try with: system.record_time(is_cuda=False)")
    return init_data_model_parallel.default_env


def record_time(is_cuda=None):
    is_cuda = is_cuda if is_cuda is not None else get_local_session().is_cuda
    if is_cuda:
        import torch
        torch.cuda.synchronize()
    return time.time()


def apply_rank_size_from_pattern(filename, rank, size, create_dir=True):
    if not re.search(r"\{rank\}", filename):
        logging.warning("Keyword `{rank}` is not found in file pattern: %s, which may cause collision in file access." % filename)
    filename = filename.replace("{size}", str(size)).replace("{rank}", str(rank))
    if create_dir and os.path.dirname(filename):
        os.makedirs(os.path.dirname(filename), exist_ok=True)
    return filename
