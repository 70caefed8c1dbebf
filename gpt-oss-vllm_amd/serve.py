"""Serving glue: launcher, readiness probe and smoke checks for an OpenAI-compatible vLLM server that runs the
MI355X NF4 MoE path through the plug-in (no edits of site-packages).

Takes over what the reference does with patches/gpt_oss_server.py (argument names and defaults :12-22, the
vLLM flags :35-46, the child process that imports a patch and then runs the API server :51-66),
patches/check_server_status.py (log-based readiness :30-58) and patches/smoke_suite.py (wait loop :10-24, one
completion :27-39, one streamed chat :42-70).  vLLM is not installed in this image: the launcher reports that and
exits with status 2 instead of starting anything.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import re
import subprocess
import sys
import time
from typing import Dict, List, Optional

DEFAULT_MODEL = "unsloth/gpt-oss-20b-unsloth-bnb-4bit"      # gpt_oss_server.py:36


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="gpt-oss NF4 server on MI355X (vLLM + mi355_nf4_moe plug-in)")
    p.add_argument("--model", default=DEFAULT_MODEL)
    p.add_argument("--host", default="0.0.0.0", help="Host to bind to")
    p.add_argument("--port", type=int, default=8000, help="Port to bind to")
    p.add_argument("--gpu-memory-utilization", type=float, default=0.90,
                   help="fraction of the 288 GB the server may take (the reference's 0.40 was sized for 24 GB)")
    p.add_argument("--max-model-len", type=int, default=256, help="Maximum model context length")
    p.add_argument("--disable-log-requests", action="store_true", help="Disable request logging")
    p.add_argument("--enforce-eager", action="store_true",
                   help="the reference always passes this; the kernels here are graph-safe, so it is optional")
    p.add_argument("--expert-parallel", type=int, default=1, metavar="N",
                   help="shard the experts over N GPUs of the node (vLLM --tensor-parallel-size N --enable-expert-parallel)")
    p.add_argument("--layout", choices=("reference", "true_bnb"), default="reference",
                   help="how checkpoint tensors are read (see Mi355Nf4MoEConfig)")
    return p


def build_vllm_args(args: argparse.Namespace) -> List[str]:
    """The api_server command line: the reference's list (gpt_oss_server.py:35-46) with the quantisation method
    swapped for the plug-in's and the bitsandbytes load format dropped."""
    out = ["--model", args.model, "--host", args.host, "--port", str(args.port), "--quantization", "mi355_nf4_moe",
           "--dtype", "float16", "--max-model-len", str(args.max_model_len),
           "--gpu-memory-utilization", str(args.gpu_memory_utilization), "--trust-remote-code"]
    if args.enforce_eager:
        out.append("--enforce-eager")
    if args.expert_parallel > 1:
        out += ["--tensor-parallel-size", str(args.expert_parallel), "--enable-expert-parallel"]
    if args.layout != "reference":
        out += ["--hf-overrides", json.dumps({"quantization_config": {"quant_method": "mi355_nf4_moe", "layout": args.layout}})]
    if args.disable_log_requests:
        out.append("--disable-log-requests")
    return out


def child_code(vllm_args: List[str]) -> str:
    """Python source the child process runs: register the plug-in, then vLLM's API server (the reference imports
    its router monkey patch at the same place, gpt_oss_server.py:59-65)."""
    return ("import sys; import gpt_oss_vllm_amd.plugin as _p; _p.register(); "
            "from vllm.entrypoints.openai import api_server as _s; "
            "sys.argv = ['api_server'] + " + repr(vllm_args) + "; _s.main()")


def main(argv: Optional[List[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    if importlib.util.find_spec("vllm") is None:
        print("vLLM is not installed in this environment: nothing was started. Install a ROCm build of vLLM, then run "
              "this launcher again; the plug-in registers itself as --quantization mi355_nf4_moe.", file=sys.stderr)
        return 2
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = os.environ.copy()
    env["PYTHONPATH"] = repo_root + (":" + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    print(f"server: http://{args.host}:{args.port}  model: {args.model}  experts over {args.expert_parallel} GPU(s)")
    try:
        return subprocess.run([sys.executable, "-c", child_code(build_vllm_args(args))], env=env).returncode
    except KeyboardInterrupt:
        return 0


# ---------------------------------------------------------------------------------------------
# readiness and smoke checks
# ---------------------------------------------------------------------------------------------
ERROR_MARKS = ("ERROR", "ValueError", "RuntimeError", "AssertionError")      # check_server_status.py:47


def log_status(log_text: str) -> Dict[str, object]:
    """What check_server_status.py:34-58 reads off the server log: {"state": "ready" | "loading" | "error" |
    "starting", ...}."""
    if "Application startup complete" in log_text:
        return {"state": "ready"}
    errors = [ln for ln in log_text.split("\n") if any(m in ln for m in ERROR_MARKS)]
    if errors:
        return {"state": "error", "lines": errors[-5:]}
    layers = re.findall(r"Converting layer\s+(\d+)/(\d+)", log_text)
    if layers:
        return {"state": "loading", "layer": int(layers[-1][0]), "of": int(layers[-1][1])}
    return {"state": "starting"}


def wait_for_server(url: str, timeout_s: float = 1800.0, poll_interval: float = 10.0) -> bool:
    """Poll GET {url}/v1/models until it answers 200 (smoke_suite.py:10-24)."""
    import requests
    deadline = time.time() + timeout_s
    while time.time() < deadline:
        try:
            if requests.get(f"{url}/v1/models", timeout=5).status_code == 200:
                return True
        except requests.exceptions.RequestException:
            pass
        time.sleep(poll_interval)
    return False


def smoke_completion(url: str, model: str = DEFAULT_MODEL) -> bool:
    """One greedy single-token completion (smoke_suite.py:27-39)."""
    import requests
    r = requests.post(f"{url}/v1/completions", json={"model": model, "prompt": "The", "max_tokens": 1, "temperature": 0.0},
                      timeout=60)
    if r.status_code != 200:
        return False
    text = r.json()["choices"][0]["text"]
    return isinstance(text, str) and len(text) > 0


def smoke_chat_stream(url: str, model: str = DEFAULT_MODEL) -> bool:
    """One streamed chat completion; true if any content delta arrives before [DONE] (smoke_suite.py:42-70)."""
    import requests
    payload = {"model": model, "messages": [{"role": "user", "content": "Say hello succinctly."}], "max_tokens": 8,
               "temperature": 0.0, "stream": True}
    with requests.post(f"{url}/v1/chat/completions", json=payload, stream=True, timeout=60) as resp:
        if resp.status_code != 200:
            return False
        for line in resp.iter_lines(decode_unicode=True):
            if not line or not line.startswith("data: "):
                continue
            data = line[len("data: "):]
            if data.strip() == "[DONE]":
                break
            try:
                if json.loads(data).get("choices", [{}])[0].get("delta", {}).get("content", ""):
                    return True
            except json.JSONDecodeError:
                continue
    return False


if __name__ == "__main__":
    sys.exit(main())
