// placeholder
#pragma once
