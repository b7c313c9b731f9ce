"""Drop-in module named ``Modules``: the reference pickles its model by reference to ``Modules.Classifier``
(main.py:322, :685), and its consumers (predict_multiway.py:111, denoise_contact.py:99) unpickle it with
``torch.load``.  Putting this repository root on ``sys.path`` (in place of the reference's ``Code/``) makes both
directions work with the MI355X-native implementation in ``matcha_amd``."""
from matcha_amd.Modules import *  # noqa: F401,F403
from matcha_amd.Modules import __all__  # noqa: F401
