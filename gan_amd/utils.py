"""Loss-dict factories of the reference (utils.py:32-53): the key strings are the contract of the metrics JSON files
and figure names.  (Figures: gan_amd/runner.py.)"""


def pix2pix_losses():
    """utils.py:32-40"""
    return {'Generator Total Loss': [],
            'Generator Loss (Primary)': [],
            'Generator Loss (Secondary)': [],
            'Discriminator Loss': []}


def cyclegan_losses():
    """utils.py:42-53"""
    return {'X->Y Generator Loss': [],
            'Y->X Generator Loss': [],
            'Total Cycle Loss': [],
            'Total X->Y Generator Loss': [],
            'Total Y->X Generator Loss': [],
            'Discriminator X Loss': [],
            'Discriminator Y Loss': []}
