"""Stand-in for the absent ``gdown`` package: lets ``oracle/gen_golden_data.py`` import the reference's
``models/dataset.py`` (module-level ``import gdown``).  Nothing here is ever called by the fixtures."""


def download(*_args: object, **_kwargs: object) -> str:
    msg = "gdown stand-in: no network in the build container"
    raise RuntimeError(msg)
