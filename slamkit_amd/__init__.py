"""slamkit_amd: the SpeechLM pre-training hot path of slamkit on MI355X."""
import os

# A data-parallel step drives four streams at once (backward, the engine's weight-gradient stream, the reducer's
# communication stream, RCCL's own); the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and two streams sharing a queue serialise behind each other's event waits - measured on one MI355X with a 1-rank
# RCCL group: 34.6 ms per Slam-358M step with 4 queues, 27.1 ms with 8. The runtime reads the
# variable when it initialises, so it is set on package import, before the first HIP call; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
