from . import seeding  # noqa
