"""Serving glue (launcher arguments, log-based readiness, smoke checks) without vLLM: the HTTP checks run against a
small in-process server that answers like an OpenAI-compatible endpoint."""
import json
import threading
from http.server import BaseHTTPRequestHandler, HTTPServer

import pytest

from gpt_oss_vllm_amd import serve


def test_launcher_arguments():
    a = serve.build_parser().parse_args([])
    v = serve.build_vllm_args(a)
    assert v[v.index("--quantization") + 1] == "mi355_nf4_moe" and "--load-format" not in v
    assert v[v.index("--max-model-len") + 1] == "256" and v[v.index("--dtype") + 1] == "float16"
    assert "--enforce-eager" not in v and "--trust-remote-code" in v
    a = serve.build_parser().parse_args(["--expert-parallel", "8", "--enforce-eager", "--disable-log-requests", "--layout", "true_bnb"])
    v = serve.build_vllm_args(a)
    assert v[v.index("--tensor-parallel-size") + 1] == "8" and "--enable-expert-parallel" in v
    assert "--enforce-eager" in v and v[-1] == "--disable-log-requests"
    assert json.loads(v[v.index("--hf-overrides") + 1])["quantization_config"]["layout"] == "true_bnb"
    code = serve.child_code(v)
    assert "_p.register()" in code and "api_server" in code
    compile(code, "<child>", "exec")


def test_launcher_without_vllm(capsys):
    import importlib.util
    if importlib.util.find_spec("vllm") is not None:
        pytest.skip("vLLM is installed here")
    assert serve.main([]) == 2
    assert "not installed" in capsys.readouterr().err


def test_log_status():
    assert serve.log_status("INFO: Application startup complete.")["state"] == "ready"
    s = serve.log_status("Converting layer 3/24\nConverting layer 4/24\n")
    assert s == {"state": "loading", "layer": 4, "of": 24}
    s = serve.log_status("Converting layer 1/24\nValueError: bad\n")
    assert s["state"] == "error" and "ValueError" in s["lines"][-1]
    assert serve.log_status("booting")["state"] == "starting"


class _Fake(BaseHTTPRequestHandler):
    def log_message(self, *a):
        pass

    def do_GET(self):
        self.send_response(200 if self.path == "/v1/models" else 404)
        self.end_headers()
        self.wfile.write(b"{}")

    def do_POST(self):
        body = json.loads(self.rfile.read(int(self.headers["Content-Length"])))
        if self.path == "/v1/completions":
            out = json.dumps({"choices": [{"text": " answer"}]}).encode()
            self.send_response(200)
            self.send_header("Content-Type", "application/json")
            self.end_headers()
            self.wfile.write(out)
        elif self.path == "/v1/chat/completions" and body.get("stream"):
            self.send_response(200)
            self.send_header("Content-Type", "text/event-stream")
            self.end_headers()
            for piece in ("", "Hel", "lo"):
                self.wfile.write(("data: " + json.dumps({"choices": [{"delta": {"content": piece}}]}) + "\n\n").encode())
            self.wfile.write(b"data: [DONE]\n\n")
        else:
            self.send_response(404)
            self.end_headers()


def test_wait_and_smoke_against_fake_endpoint():
    srv = HTTPServer(("127.0.0.1", 0), _Fake)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    url = f"http://127.0.0.1:{srv.server_address[1]}"
    try:
        assert serve.wait_for_server(url, timeout_s=5, poll_interval=0.1)
        assert serve.smoke_completion(url)
        assert serve.smoke_chat_stream(url)
        assert not serve.wait_for_server("http://127.0.0.1:9", timeout_s=0.5, poll_interval=0.1)
    finally:
        srv.shutdown()
