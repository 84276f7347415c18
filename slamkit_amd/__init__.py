"""slamkit_amd: the SpeechLM pre-training hot path of slamkit on MI355X."""
import os

# A data-parallel step drives four streams at once (backward, the engine's weight-gradient stream, the reducer's
# communication stream, RCCL's own); the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and two streams sharing a queue serialise behind each other's event waits - measured on one MI355X with a 1-rank
# RCCL group: 34.6 ms per Slam-358M step with 4 queues, 27.1 ms with 8. The runtime reads the
# variable when it initialises, so it is set on package import, before the first HIP call; an explicit setting wins.
HW_QUEUES_SET_LATE = False
if "GPU_MAX_HW_QUEUES" not in os.environ:
    try:
        import torch as _torch
        HW_QUEUES_SET_LATE = bool(_torch.cuda.is_initialized())  # HIP already up: the setting below no longer takes effect
    except Exception:  # noqa: BLE001
        pass
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


def check_hw_queues(min_queues: int = 8) -> None:
    """Fail loudly before a data-parallel run drives four streams over too few hardware queues (the slow path is silent:
    +8 ms per Slam-358M step with the default 4). SLAM_ALLOW_FEW_HW_QUEUES=1 downgrades the error to a warning."""
    n = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    if n >= min_queues and not HW_QUEUES_SET_LATE:
        return
    msg = (f"GPU_MAX_HW_QUEUES={n}{' (set after HIP was initialised: not in effect)' if HW_QUEUES_SET_LATE else ''}: the data-parallel "
           f"step needs >= {min_queues} hardware queues (backward, weight-gradient, communication and RCCL streams); export "
           f"GPU_MAX_HW_QUEUES={min_queues} before the first HIP call, or import slamkit_amd before torch.cuda is initialised")
    if os.environ.get("SLAM_ALLOW_FEW_HW_QUEUES", "0") == "1":
        import logging
        logging.getLogger(__name__).warning(msg)
    else:
        raise RuntimeError(msg)
