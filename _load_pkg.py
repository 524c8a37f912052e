"""Load the product package by path: its directory name (`llama.cpp_test_amd`) contains a dot, so it cannot be
imported by name.  Returns the module, registered as `llama_cpp_test_amd`."""
import importlib.util
import os
import sys

_NAME = "llama_cpp_test_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llama.cpp_test_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
