import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _glnn_options_follow_monkeypatch(monkeypatch):
    """libglnn_hip.so reads its GLNN_* switches once (glnn::Options); tests that flip one through monkeypatch.setenv / delenv between
    two runs need the library to re-read them: the two methods are wrapped to call glnn_reload_options(), and the switches are restored
    (environment first, then a reload) when the test ends."""
    def reload():
        lib = sys.modules.get("glnn_amd._lib")
        h = getattr(lib, "_lib", None) if lib else None
        if h is not None:
            h.glnn_reload_options()

    touched = []
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set0(name, value, prepend)
        if name.startswith("GLNN_"):
            touched.append(name)
            reload()

    def delenv(name, raising=True):
        del0(name, raising)
        if name.startswith("GLNN_"):
            touched.append(name)
            reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    if touched:
        monkeypatch.undo()
        reload()
