"""Exception types of numpywren_amd.

The first group mirrors the reference's numpywren/exceptions.py (same class names) so user code
catching them keeps working; the second group is specific to the HIP backend.
"""


class LambdaPackParsingException(Exception):
    pass


class LambdaPackTypeException(Exception):
    pass


class LambdaPackBackendGenerationException(Exception):
    pass


class LambdaPackTimeoutException(Exception):
    pass


class HipExtensionError(RuntimeError):
    """libnpw_hip.so is missing / cannot be loaded, or no gfx950 device is visible."""


class NpwHipError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__(f"libnpw_hip error {code}: {message}")
        self.code = code
        self.message = message
