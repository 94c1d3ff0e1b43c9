class UNet2DConditionLoadersMixin:
    pass


class FromSingleFileMixin:
    pass


class IPAdapterMixin:
    pass


class StableDiffusionXLLoraLoaderMixin:
    pass


class TextualInversionLoaderMixin:
    pass
