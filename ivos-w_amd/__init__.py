"""ivos_w_amd — MI355X-native hot path of IVOS-W (quality-assessment CNN + Double-DQN agent).

Host side mirrors the reference class surfaces (``models.agent``, ``models.assessment``,
``models.momory_pool``, ``datasets.agent_dataset``, ``utils.utils_agent``, ``utils.misc``);
all arithmetic runs in ``libivosw_hip.so`` (hand-written HIP for gfx950) behind the C ABI
declared in ``include/ivosw.h``.  There is no CPU fallback: calling a compute entry point
without the library, or with CPU tensors, raises.
"""
__version__ = "0.1.0"
