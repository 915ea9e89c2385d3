"""MI355X-native drop-in for the `vllm_flash_attn` package (same names and signatures as
/root/reference/vllm_flash_attn/__init__.py:1-11)."""
__version__ = "2.5.8.post3"

from vllm_flash_attn.flash_attn_interface import (  # noqa: F401
    enable_cpu_fp32,
    flash_attn_func,
    flash_attn_kvpacked_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
    flash_attn_with_kvcache,
)
