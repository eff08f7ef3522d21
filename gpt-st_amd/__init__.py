"""gpt-st_amd — MI355X-native GPT-ST masked-autoencoder pretraining hot path.

Host side is PyTorch-ROCm Python mirroring the reference's ``GPTST_Model`` /
``Run.py -mode pretrain`` interface; all compute on the path runs in hand-written
HIP kernels (gfx950) behind the C ABI declared in ``include/gptst_hip.h``.
There is no CPU fallback: importing the compute modules without the built
``libgptst_hip.so`` raises.
"""
__version__ = "0.1.0"
