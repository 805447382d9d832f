"""Reference-side binding stubs a maintainer of the reference would add.  Nothing in the product imports this
package: ``bagel_amd/integration/flash_attn`` is put on ``sys.path`` IN FRONT of the reference tree to replace its one by-name
native import (qwen2_navit.py:24, siglip_navit.py:14) while the rest of the reference stays eager PyTorch."""
