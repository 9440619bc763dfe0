"""Mirror of lib/StarkError.ts:3-13."""


class StarkError(Exception):
    def __init__(self, message, cause=None):
        if cause is not None:
            message = f'{message}: {cause}'
        super().__init__(message)
        self.cause = cause
