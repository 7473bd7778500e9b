"""Test-infrastructure stub: minimal `omegaconf` so /root/reference imports in this container.

Only used by oracle/ref_import.py (oracle validation + golden generation). Not product code.
The reference touches omegaconf at diffbir/model/unet.py:428, controlnet.py:92 (ListConfig type
check) and inference/loop.py:8 (OmegaConf.load)."""
import yaml


class ListConfig(list):
    pass


class DictConfig(dict):
    pass


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return yaml.safe_load(f)

    @staticmethod
    def to_container(cfg, resolve=True):
        return cfg
