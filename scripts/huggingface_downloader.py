"""Adapter / model download service of the LoRA sidecar: ``POST /model/download`` with
``{"model_id", "local_dir", "token"}`` fetches a Hugging Face repository into
``$LORA_DOWNLOAD_BASE_DIR/<local_dir>`` (the reference ships the same service under this path:
/root/reference/scripts/huggingface_downloader.py:1-48).  The implementation lives in
``vllm_production_stack_amd.sidecar``; this file keeps the reference's entry point working:

    python scripts/huggingface_downloader.py [--port 8000]
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_production_stack_amd.sidecar import main  # noqa: E402

if __name__ == "__main__":
    main()
