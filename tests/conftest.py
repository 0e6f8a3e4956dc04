import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
ASSETS = os.path.join(ROOT, "assets")
SCENES = ("simple_sphere", "multisphere", "suzanne", "bunny")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(ASSETS, "scene", name + ".scene")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def rtm():
    from rtm_loader import load_package
    return load_package()


@pytest.fixture(scope="session")
def rtm_gpu(rtm):
    rtm.init(0)  # raises (loudly) when no GPU is visible: there is no CPU fallback
    return rtm


_oracle_cache = {}
_rtm_cache = {}


@pytest.fixture(scope="session")
def oracle_scene(orc):
    def get(name):
        if name not in _oracle_cache:
            _oracle_cache[name] = orc.OracleScene(path=scene_path(name), base_dir=ASSETS)
        return _oracle_cache[name]
    return get


@pytest.fixture(scope="session")
def rtm_scene(rtm):
    def get(name):
        if name not in _rtm_cache:
            _rtm_cache[name] = rtm.Scene(path=scene_path(name), base_dir=ASSETS)
        return _rtm_cache[name]
    return get
